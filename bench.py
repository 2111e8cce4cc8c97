#!/usr/bin/env python3
"""Headline benchmark: MC samples/sec of the SPDE Matérn sampler (BASELINE.json config 2:
PDESampler on cube_tet refined 5x -> 595 968 DoF, single Monte Carlo level) on N MI355X.

A step = Sample + Eval of `--streams` batches of `--batch` realizations, white noise drawn on the device so
all inputs are resident in HBM when the timed region starts.  Contract: W untimed warm-up steps,
then exactly K steps bracketed by barrier + torch.cuda.synchronize(); MAX over ranks; rank 0
prints ONE JSON line.  The timed region carries no instrumentation; the roofline figures are taken in
separate short passes after it.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL),
every rank owns a full replica of the operators and its own realizations (weak scaling, no data-path
collective); the one exchange of a sample farm - the SUM all-reduce of the MLMC accumulators - is exercised
through the library's own communicator (pmc_comm_* / pmc_allreduce_sum_f64, RCCL) under extra.mlmc_farm.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def build_problem(nref, extra_coarse=True):
    from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json
    mesh = mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json"))
    h = build_hierarchy(mesh, nref)
    # one Monte Carlo level (config 2); the coarser refinement levels only deepen the V-cycle
    return build_sampler_problem(h, corlen=0.1, lognormal=False, n_mc_levels=1)


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota and by
    PMC_CPU_CORES; a 1-GPU box grants a 16-core share of the host, so that is the default cap."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:   # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("PMC_CPU_CORES", "16"))))


def cpu_baseline(problem, seed, nsamples_per_core=12):
    """Reference algorithm restated in C (oracle/c/pmc_ref.c), farmed over the host cores: a bounded
    sample of the same workload."""
    from oracle.cport import CPort
    from oracle.rng_oracle import normal_fill
    cores = host_cores()
    cp = CPort(problem)
    ns = nsamples_per_core * cores
    n = problem.levels[0].n_s
    xi = np.stack([normal_fill(n, seed, i, 0) for i in range(ns)])
    rhs = cp.rhs(0, 0, xi)
    cp.solve(0, rhs[:cores], nthreads=cores)          # warm-up (page in, thread pool)
    t0 = time.perf_counter()
    _, iters = cp.solve(0, rhs, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": ns / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{ns} realizations ({nsamples_per_core} per core), MINRES(300,1e-6)+BJ[symGS x3 | V-cycle], "
                      f"mean {float(np.mean(np.abs(iters))):.1f} iterations, {dt:.1f} s wall"}


def solver_bytes_per_iteration(problem, nb):
    """ALGORITHMIC bytes one MINRES iteration of a batch of nb realizations moves on level 0 (every
    operand once; matrices 12 B per stored nonzero + 4 B per row, vectors 8 nb B per row touched)."""
    import scipy.sparse as sp
    L = problem.levels
    V = 8.0 * nb

    def mat(nnz, nrows):
        return 12.0 * nnz + 4.0 * nrows

    n_u, n_s = L[0].n_u, L[0].n_s
    n = n_u + n_s
    total = mat(L[0].nnz, n) + V * 2 * n                      # K5: q = A u (the dot takes u from the gathers)
    total += V * 4 * n                                         # v_new = c0 q + c1 v1 + c2 v0
    total += mat(L[0].M.nnz, n_u) + 8.0 * n_u + V * 2 * n_u    # M-block: one-pass degree-2 polynomial
    total += V * 6 * n_s                                       # w / x update on the s-block
    # V-cycle on the Schur block, level by level until the first level handled by the LDS tail (<= ~6k rows) / last level
    for lv in range(len(L)):
        ns_l = L[lv].n_s
        if ns_l <= 6000 or lv == len(L) - 1:
            total += V * 2 * ns_l                              # tail: r in, x out (matrices of the tail levels stay in L2)
            break
        B = L[lv].B.tocsr()
        nnzS = ((abs(B) @ abs(B).T) + sp.identity(ns_l)).nnz   # pattern of aW + B diag(M)^-1 B^T
        nc = L[lv + 1].n_s
        total += mat(nnzS, ns_l) + 8.0 * ns_l + V * 2 * ns_l           # pre-smoothing (one pass)
        total += mat(nnzS, ns_l) + V * (3 * ns_l + nc)                 # residual with the restriction fused in (octree P)
        total += mat(nnzS, ns_l) + V * (2 * ns_l + nc)                 # res - (S P) xc   (S P has the pattern of S)
        total += mat(nnzS, ns_l) + 12.0 * ns_l + V * (4 * ns_l + nc)   # post-smoothing + coarse correction + dot
    return total


def traffic_entry(key):
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(tfile)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:   # noqa: BLE001
        return None


class SamplerFarm:
    """`streams` independent batches in flight on one GPU: one context + sampler + buffers each."""

    def __init__(self, problem, dev, seed, nb, streams, world=1, rank=0):
        from parelagmc_amd import capi
        self.nb, self.ns, self.world, self.rank = nb, streams, world, rank
        self.n = problem.levels[0].n_s
        self.lanes = []
        for _ in range(streams):
            c = capi.Context(dev, seed=seed)
            c.seed(seed, nparts=world, mypart=rank)
            self.lanes.append((c, capi.PDESampler(c, problem), c.empty(nb * self.n), c.empty(nb * self.n)))

    def one_batch(self, lane, batch_index):
        c, sm, xi_d, s_d = self.lanes[lane]
        first = batch_index * self.nb          # realization ids are never repeated
        sm.Sample(0, first_id=first, nbatch=self.nb, out=xi_d)
        return sm.Eval(0, xi_d, xi_level=0, s_out=s_d, return_stats=True)[1]

    def step(self, i):
        """`streams` batches of nb realizations in flight at once.  Batch ids are LOCAL to the rank: the generator is split
        over the ranks (pmc_rng_seed nparts / mypart = NormalDistributionSampler::Split), so no two ranks share a realization"""
        res = [None] * self.ns
        base = i * self.ns

        def work(lane):
            res[lane] = self.one_batch(lane, base + lane)
        if self.ns == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(self.ns)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return [t for r in res for t in r]

    def close(self):
        for c, sm, _, _ in self.lanes:
            sm.close()
            c.close()
        self.lanes = []


def check_stats(stats, what):
    bad = [t for t in stats if t[1] != 1]
    if bad:
        print(f"bench: {len(bad)} of {len(stats)} realizations did not converge in {what}: {bad[:4]}", file=sys.stderr)
        sys.exit(3)


def operator_roofline(farm, problem, nb, refine, next_batch, solver_bytes, iters_total, dt):
    """Roofline block of the dominant kernel, the block saddle-point SpMM K5, measured after the timed region:
    `achieved` is the instantiation the MINRES loop launches (fused <u, Au>), every launch of three solo batches of lane 0
    bracketed by HIP events on the lane's stream; `isolated` = the plain product launched back to back (operator and
    vectors then stay in the Infinity Cache at r = 5)."""
    ctx, smp = farm.lanes[0][0], farm.lanes[0][1]
    L = problem.levels[0]
    smp.set_operator_timing(True)
    smp.operator_time()
    for j in range(3):
        check_stats(farm.one_batch(0, next_batch + j), "the in-loop operator pass")
    gap_ms = smp.operator_event_overhead()
    solo_ms, solo_launches = smp.operator_time()
    smp.set_operator_timing(False)
    x = ctx.array(np.random.default_rng(0).standard_normal(nb * (L.n_u + L.n_s)))
    _, k_ms, k_bytes = smp.Mult(0, x, repeat=50)
    x1 = ctx.array(np.random.default_rng(0).standard_normal(L.n_u + L.n_s))
    _, k1_ms, k1_bytes = smp.Mult(0, x1, repeat=50)
    # (hipGraph replay, opts.use_graph, cannot be bracketed by events: fall back to the isolated launches then)
    # A bracket = event record, launch, event record: it contains what an event record costs on that stream (~6 us), measured
    # by the empty bracket behind every timed launch.  Net of it the live number agrees with the rocprofv3 average of the
    # same kernel (profiles/rNN_bench_s1_kernel_stats.csv: 56.5 vs 57.0 us at r = 5); the raw bracket is reported beside it.
    raw_ms = solo_ms / solo_launches if solo_launches > 0 else k_ms
    gap = gap_ms / solo_launches if solo_launches > 0 else 0.0
    loop_ms = raw_ms - gap
    ach = k_bytes / (loop_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"pmc::sell_spmm_kernel<{nb}, false, 0, true, 1> (block operator K5 as launched by the "
                                     "MINRES loop: fused <u, Au>, one lane alone on the GPU)",
           "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
           "traffic": traffic_entry(f"r{refine}_nb{nb}_inloop"),
           "bytes_per_launch": k_bytes, "avg_kernel_ms": loop_ms, "launches": solo_launches,
           "raw_event_bracket_ms": raw_ms, "event_overhead_ms": gap,
           "frac_of_raw_bracket": k_bytes / (raw_ms * 1e-3) / 1e9 / PEAK_GBS,
           "isolated": {"kernel": f"pmc::sell_spmm_kernel<{nb}, false, 0, false, 2> launched back to back",
                        "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "frac": k_bytes / (k_ms * 1e-3) / 1e9 / PEAK_GBS,
                        "avg_kernel_ms": k_ms, "traffic": traffic_entry(f"r{refine}_nb{nb}")},
           "spmv_nb1": {"achieved": k1_bytes / (k1_ms * 1e-3) / 1e9, "bytes_per_launch": k1_bytes, "avg_kernel_ms": k1_ms,
                        "frac": k1_bytes / (k1_ms * 1e-3) / 1e9 / PEAK_GBS},
           # the whole solver: algorithmic bytes of every kernel of a MINRES iteration x batch-iterations executed in
           # the timed region / its wall time
           "solver": {"bytes_per_iteration": solver_bytes, "batch_iterations": iters_total,
                      "achieved": solver_bytes * iters_total / dt / 1e9,
                      "frac": solver_bytes * iters_total / dt / 1e9 / PEAK_GBS}}
    return out


def mlmc_config3(seed, lanes=4, opts=None, farm=None, batch=256):
    """Secondary figure (BASELINE config 3): MLMC_Manager::InitRun with the SPDE sampler + Darcy QoI on cube_hex
    64^3 / 32^3 / 16^3, fixed sample counts, `lanes` concurrent streams.  Reported under "extra", never as `value`.
    farm = (world, rank, comm_ctx, device): the realizations of every level are sharded over the ranks and the accumulators
    are all-reduced through the library's own RCCL communicator of comm_ctx (MLMC_Manager::SetFarm with reduce == NULL)."""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
    sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
    dev = farm[3] if farm else 0
    # farm: the manager's primary context is the one that carries the RCCL communicator
    ctxs = ([farm[2]] if farm else []) + [capi.Context(dev, seed=seed) for _ in range(lanes - (1 if farm else 0))]
    sm = [capi.PDESampler(c, sp, opts) for c in ctxs]
    dr = [capi.DarcySolver(c, dp, opts) for c in ctxs]
    # batch = upper limit of realizations per plugin call; per level the manager hands over what the plugins prefer
    # (pmc_sampler_batch_width: 16 at a time on the bandwidth-bound 64^3 level, 64 on 32^3, 256 - eight column groups of 32
    # in one launch - on 16^3), cut so that every lane gets a share
    mgr = host_api.MLMCManager(3, sampler=sm[0], solver=dr[0], wall_time=True, batch=batch)
    for i in range(1, lanes):
        mgr.add_lane(sm[i], dr[i])
    world = 1
    if farm:
        world, rank = farm[0], farm[1]
        mgr.set_farm(world, rank, None)              # reduce == NULL -> pmc_allreduce_sum_f64 (RCCL) of ctxs[0]
    mgr.InitRun([64 * world, 256 * world, 1024 * world])    # warm-up: allocations at the widths of the timed round
    mgr.Reset()
    ns = [64 * world, 256 * world, 1024 * world]
    t0 = time.perf_counter()
    r = mgr.InitRun(ns)
    dt = time.perf_counter() - t0
    out = {"workload": "MLMC Darcy + SPDE sampler, cube_hex 64^3/32^3/16^3 (1 060 864 / 134 144 / 17 152 DoF), lognormal, "
                       f"eff_perm QoI, InitRun {ns}, {lanes} lanes, realizations per launch per level "
                       f"{[sm[0].BatchWidth(l) for l in range(3)]}" + (f", sharded over {world} ranks" if farm else ""),
           "realizations_per_s": sum(ns) / dt, "seconds": dt, "estimate": r["estimate"],
           "nsamples_after_allreduce": [int(x) for x in r["nsamples"]],
           "seconds_per_sample_per_level": [float(x) for x in r["cost"]], "varY": [float(x) for x in r["varY"]]}
    mgr.close()
    for c in ctxs:
        c.close()
    return out, (sp, dp)


def darcy_cpu_baseline(sp, dp, seed, per_core=(1, 2, 8)):
    """CPU column of config 3: the reference's per-realization work restated in C (oracle/c/pmc_ref.c) - sampler solve,
    then re-assemble M(k), REBUILD the preconditioner and solve the Darcy system, as src/DarcySolver.cpp:472-649 does for
    every sample - on a bounded sample per level, farmed over the host cores.  An MLMC realization of level l < L-1 is a
    PAIR: the same work on level l+1 as well (src/MLMC_Manager.cpp:144-173)."""
    from oracle.cport import CPort, DarcyCPort
    from oracle.rng_oracle import normal_fill
    cores = host_cores()
    cs, cd = CPort(sp), DarcyCPort(dp)
    single = []
    for lvl in range(3):
        ns = per_core[lvl] * cores
        n = sp.levels[lvl].n_s
        xi = np.stack([normal_fill(n, seed, i, lvl) for i in range(ns)])
        t0 = time.perf_counter()
        sol, _ = cs.solve(lvl, cs.rhs(lvl, lvl, xi), nthreads=cores)
        kf = np.exp(sol[:, sp.levels[lvl].n_u:])
        _, it = cd.solve(lvl, kf, nthreads=cores)
        single.append((time.perf_counter() - t0) / ns)
        if np.any(it <= 0):
            raise RuntimeError(f"CPU Darcy port did not converge on level {lvl}")
    pair = [single[l] + (single[l + 1] if l + 1 < 3 else 0.0) for l in range(3)]
    w = np.array([64, 256, 1024], float)
    return {"kind": "port", "cores": cores, "seconds_per_realization_per_level": pair,
            "realizations_per_s": float(w.sum() / (w * np.array(pair)).sum()),
            "sample": f"{[p * cores for p in per_core]} sampler + Darcy solves on levels 0..2 (Darcy: per-sample M(k), elimination, "
                      "Schur hierarchy refresh, MINRES); a realization of level l < 2 = the pair (l, l+1); rate = the "
                      "[64, 256, 1024] round at these costs"}


def spawn_ranks(n):
    """`python bench.py --gpus N` (N > 1) started without torchrun: the parent - which never imports torch.cuda and never
    creates a pmc_ctx - starts N fresh child processes of this script, one rank per GPU, with the torch.distributed
    environment set (as the reference starts its ranks under mpirun, examples/MLMC.cpp:43-50); rank 0 prints the JSON line
    on the inherited stdout, the other ranks' stdout goes to stderr; exit code = first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                if c != 0 and rc == 0:
                    rc = c
                    for q in live:       # a failed rank leaves the others waiting in a collective: end them
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--streams", type=int, default=4,
                    help="independent batches in flight per GPU (one context + host thread each): the launch-latency-"
                         "bound coarse-level kernels of one batch overlap the bandwidth-bound kernels of the other")
    ap.add_argument("--refine", type=int, default=5, help="uniform refinements of cube_tet (5 -> 595 968 DoF)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mlmc", action="store_true", help="skip the secondary config-3 (MLMC Darcy+SPDE) figure")
    ap.add_argument("--no-r6", action="store_true", help="skip the HBM-bound point (cube_tet r=6, 4.74 M DoF) under extra.r6")
    ap.add_argument("--seed", type=int, default=20261003)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself and never touches the GPU
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench: --gpus {args.gpus} but WORLD_SIZE {world}: the launcher must start exactly --gpus ranks", file=sys.stderr)
        sys.exit(2)
    ndev = max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # backend nccl = RCCL over xGMI.  gloo only rehearses the multi-rank code path on a box with fewer GPUs than ranks
        # (ranks then share devices round-robin; RCCL refuses two ranks on one device): chosen by PMC_BENCH_BACKEND, or by
        # itself when the box has fewer devices than ranks - the line then says so ("devices" < "n_gpus")
        backend = os.environ.get("PMC_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend, rank=rank, world_size=world)

    from parelagmc_amd import capi

    problem = build_problem(args.refine)
    L = problem.levels[0]
    dev = (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0
    red_dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    nb, ns = args.batch, max(1, args.streams)
    farm = SamplerFarm(problem, dev, args.seed, nb, ns, world, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        farm.step(i)
    barrier()
    t0 = time.perf_counter()
    stats = []
    for i in range(args.steps):
        stats += farm.step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    check_stats(stats, "the timed region")
    iters = [t[0] for t in stats]
    # batch-iterations executed: a batch runs until its slowest column has converged
    batch_iters = float(sum(max(iters[b:b + nb]) for b in range(0, len(iters), nb)))
    acc = np.array([float(np.sum(iters)), float(len(iters)), batch_iters])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        a = torch.from_numpy(acc).to(red_dev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        acc = a.cpu().numpy()
    total_samples = args.steps * nb * ns * world
    value = total_samples / dt
    # what every rank did in the timed region: local realization ids [first, last] of a generator split nparts = world,
    # mypart = rank, i.e. the global ids first * world + rank, ..., last * world + rank - disjoint by construction
    mine = {"rank": rank, "device": dev, "samples": args.steps * nb * ns, "seconds": dt_local,
            "global_ids": [args.warmup * ns * nb * world + rank, ((args.warmup + args.steps) * ns * nb - 1) * world + rank],
            "id_stride": world}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    out = None
    if rank == 0:
        sbytes = solver_bytes_per_iteration(problem, nb)
        next_batch = (args.warmup + args.steps + 1) * ns
        out = {
            "metric": "MC samples/sec (SPDE field + Darcy QoI) at stated DoF; SpMV HBM GB/s vs roofline",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"PDESampler cube_tet r={args.refine}, {L.n_u + L.n_s} DoF "
                                   f"(n_s={L.n_s}, n_u={L.n_u}, nnz(A)={L.nnz}), 1 MC level, corlen 0.1: the SPDE field only, as "
                                   f"BASELINE config 2 is (the Darcy QoI leg is measured on config 3 under extra.mlmc_config3), "
                                   f"MINRES 300/1e-6/1e-12, {ns} x {nb} realizations per step, all converged",
                       "mean_minres_iterations": acc[0] / max(acc[1], 1.0), "batch": nb, "streams": ns,
                       "parallelism": f"sample-farm x{world}"},
            "devices": min(world, ndev), "ranks": per_rank,
            "roofline": operator_roofline(farm, problem, nb, args.refine, next_batch, sbytes, acc[2], dt),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(problem, args.seed)
    farm.close()
    extra = {}
    if not args.no_mlmc:
        try:
            if world == 1:
                m, probs = mlmc_config3(args.seed)
                if not args.no_cpu_baseline:
                    try:
                        m["cpu_baseline"] = darcy_cpu_baseline(probs[0], probs[1], args.seed)
                    except Exception as e:   # noqa: BLE001
                        m["cpu_baseline"] = {"error": repr(e)}
                extra["mlmc_config3"] = m
            else:
                # every rank takes part: sharded InitRun, accumulators summed by pmc_allreduce_sum_f64 (RCCL).  The
                # communicator is set up and tried FIRST, and the ranks agree (torch.distributed) on whether it works, so
                # that no rank enters the farm alone
                uid = [None]
                c0 = capi.Context(dev, seed=args.seed)
                if rank == 0:
                    uid[0] = c0.comm_unique_id()
                dist.broadcast_object_list(uid, src=0)
                ok, why = 1, ""
                try:
                    c0.comm_init(uid[0], world, rank)
                    probe = c0.allreduce_sum(np.array([1.0, float(rank)]))
                    ok = 1 if probe[0] == float(world) else 0
                except Exception as e:   # noqa: BLE001
                    ok, why = 0, repr(e)
                flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    m, _ = mlmc_config3(args.seed, farm=(world, rank, c0, dev))
                    m["ranks_in_rccl_communicator"] = world
                    if rank == 0:
                        extra["mlmc_farm"] = m
                else:
                    c0.close()
                    if rank == 0:
                        extra["mlmc_farm"] = {"error": "RCCL communicator of the library not available on every rank: " + why}
        except Exception as e:   # noqa: BLE001 - the secondary figure must never cost the headline line
            extra["mlmc_config3" if world == 1 else "mlmc_farm"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_r6 and args.refine != 6:
        try:
            p6 = build_problem(6)
            f6 = SamplerFarm(p6, dev, args.seed, nb, ns)
            for i in range(2):
                f6.step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st6 = []
            steps6 = 8
            for i in range(steps6):
                st6 += f6.step(2 + i)
            torch.cuda.synchronize()
            dt6 = time.perf_counter() - t0
            check_stats(st6, "the r=6 run")
            it6 = [t[0] for t in st6]
            bi6 = float(sum(max(it6[b:b + nb]) for b in range(0, len(it6), nb)))
            L6 = p6.levels[0]
            extra["r6"] = {"workload": f"PDESampler cube_tet r=6, {L6.n_u + L6.n_s} DoF (nnz(A)={L6.nnz}): operator + vectors "
                                       "exceed the 256 MiB Infinity Cache, every kernel is HBM-bound",
                           "value": steps6 * nb * ns / dt6, "unit": "samples/s", "steps": steps6,
                           "mean_minres_iterations": float(np.mean(it6)),
                           "roofline": operator_roofline(f6, p6, nb, 6, (2 + steps6 + 1) * ns, solver_bytes_per_iteration(p6, nb),
                                                         bi6, dt6)}
            f6.close()
        except Exception as e:   # noqa: BLE001
            extra["r6"] = {"error": repr(e)}
    if rank == 0 and extra:
        out["extra"] = extra
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
