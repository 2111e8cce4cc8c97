"""One rank of the device sample farm (started twice by tests/test_gpu_round3.py, both ranks on device 0): real
PDESampler + DarcySolver plugins with two lanes each, MLMC_Manager::SetFarm with a gloo SUM all-reduce as the reduction,
rank-sharded per-sample logs.  usage: farm_worker.py <out_dir>  (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 1)
    sp = build_sampler_problem(h, corlen=0.1, lognormal=True)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1])
    o = capi.solver_opts(rel_tol=1e-12, abs_tol=1e-14)
    ctxs = [capi.Context(0, seed=20261003) for _ in range(2)]
    sm = [capi.PDESampler(c, sp, o) for c in ctxs]
    dr = [capi.DarcySolver(c, dp, o) for c in ctxs]
    nred = [0]

    def reduce(buf):
        nred[0] += 1
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)     # shares memory with the C buffer

    mgr = host_api.MLMCManager(2, sampler=sm[0], solver=dr[0], wall_time=False, eps2=1e-3,
                               log_file=os.path.join(out_dir, "MLMC.dat"))
    mgr.add_lane(sm[1], dr[1])
    mgr.set_farm(world, rank, reduce)
    r1 = mgr.InitRun([19, 37])
    r2 = mgr.InitRun([6, 0])
    t = [mgr.phase_times(l) for l in range(2)]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), sums1=r1["sums"], sums=r2["sums"], nsamples=r2["nsamples"],
             missing=r2["missing"], varY=r2["varY"], estimate=r2["estimate"], reductions=nred[0],
             local_realizations=[t[0]["sampler_realizations"], t[1]["sampler_realizations"]])
    mgr.close()
    for x in dr + sm:
        x.close()
    for c in ctxs:
        c.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
