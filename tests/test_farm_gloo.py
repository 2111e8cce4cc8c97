"""Sample farm over 2 ranks (gloo, CPU): sharded InitRun + all-reduced sums == serial run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parelagmc_amd import host_api
    from test_mlmc_host import SyntheticPlugin

    def reduce(buf):
        t = torch.from_numpy(buf)        # shares memory with the C buffer
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=4, eps2=1e-3)
    mgr.set_farm(world, rank, reduce)
    r = mgr.InitRun([9, 14, 23])
    r2 = mgr.InitRun([5, 0, 7])
    n_eval_local = len(pl.calls)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), sums=r2["sums"], nsamples=r2["nsamples"], missing=r2["missing"],
             varY=r2["varY"], n_eval=n_eval_local, first_sums=r["sums"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_farm_matches_serial(tmp_path):
    from parelagmc_amd import host_api
    from test_mlmc_host import SyntheticPlugin
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    pl = SyntheticPlugin(3)
    mgr = host_api.MLMCManager(3, callbacks=pl.callbacks(), wall_time=False, batch=4, eps2=1e-3)
    s1 = mgr.InitRun([9, 14, 23])
    s2 = mgr.InitRun([5, 0, 7])
    for r in res:
        assert np.allclose(r["first_sums"], s1["sums"], rtol=1e-12, atol=1e-13)
        assert np.allclose(r["sums"], s2["sums"], rtol=1e-12, atol=1e-13)       # same samples, different order
        assert list(r["nsamples"]) == [14, 14, 30]
        assert list(r["missing"]) == list(s2["missing"])                       # all ranks agree on the allocation
        assert np.allclose(r["varY"], s2["varY"], rtol=1e-10)
    # the work was actually split: each rank did part of the evaluations, together all of them
    assert res[0]["n_eval"] + res[1]["n_eval"] == len(pl.calls)
    assert 0 < res[0]["n_eval"] < len(pl.calls)
    mgr.close()


def _worker8(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parelagmc_amd import host_api
    from test_mlmc_host import SyntheticPlugin
    nred = [0]

    def reduce(buf):
        nred[0] += 1
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)

    pl = SyntheticPlugin(3)
    drawn = []
    inner = pl.sample

    def sample(level, first_id, nbatch):
        drawn.append((level, int(first_id), int(nbatch)))
        return inner(level, first_id, nbatch)
    cb = pl.callbacks()
    cb["sample"] = sample
    mgr = host_api.MLMCManager(3, callbacks=cb, wall_time=False, batch=256, eps2=1e-3)
    mgr.set_farm(world, rank, reduce)
    r = mgr.InitRun([64 * world, 256 * world, 1024 * world])
    ms, n = mgr.farm_times()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), sums=r["sums"], nsamples=r["nsamples"], drawn=np.array(drawn),
             nred=nred[0], farm_reductions=n, farm_ms=ms)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_farm_rehearsal(tmp_path):
    """What `bench.py --gpus 8` asks of the manager (extra.mlmc_farm), rehearsed with EIGHT gloo ranks on the CPU (a GPU box
    admits at most six processes on its card): InitRun [512, 2048, 8192] sharded over 8 ranks, every realization id drawn
    exactly once, ONE reduction per round on every rank, the reduced counts and sums identical on all ranks and equal to
    the serial manager's (the reference's manager is serial: /root/reference/src/MLMC_Manager.hpp:24, :113-179)."""
    from parelagmc_amd import host_api
    from test_mlmc_host import SyntheticPlugin
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    ns = [64 * world, 256 * world, 1024 * world]
    assert ns == [512, 2048, 8192]
    for r in res:
        assert list(r["nsamples"]) == ns
        assert int(r["nred"]) == 1 and int(r["farm_reductions"]) == 1 and float(r["farm_ms"]) >= 0.0
        assert np.allclose(r["sums"], res[0]["sums"], rtol=0, atol=0)            # the all-reduce gives every rank the same table
    for lvl in range(3):
        ids = []
        for r in res:
            for l, first, nb in r["drawn"]:
                if l == lvl:
                    ids += list(range(first, first + nb))
        assert sorted(ids) == list(range(ns[lvl])), lvl                           # disjoint and complete
        share = [sum(nb for l, _, nb in r["drawn"] if l == lvl) for r in res]
        assert max(share) - min(share) <= ns[lvl] // world // 2 + 256 and min(share) > 0   # every rank got a share
    mgr = host_api.MLMCManager(3, callbacks=SyntheticPlugin(3).callbacks(), wall_time=False, batch=256, eps2=1e-3)
    s = mgr.InitRun(ns)
    assert np.allclose(res[0]["sums"], s["sums"], rtol=1e-11, atol=1e-12)
