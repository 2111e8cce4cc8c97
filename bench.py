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
    Z = float(z_bytes()) * nb      # a preconditioned vector z (fp32 storage unless the library is a -DPMC_Z64 build)
    F = 4.0 * nb                   # an fp32 intermediate of the V-cycle (iterate, residual)

    def mat(nnz, nrows):
        return 12.0 * nnz + 4.0 * nrows

    n_u, n_s = L[0].n_u, L[0].n_s
    n = n_u + n_s
    total = mat(L[0].nnz, n) + (Z + V) * n                    # K5: q = A u (u is a z; the dot takes u from the gathers)
    total += V * 4 * n                                         # v_new = c0 q + c1 v1 + c2 v0
    total += mat(L[0].M.nnz, n_u) + 8.0 * n_u + (V + Z) * n_u  # M-block: one-pass degree-2 polynomial, r in, z out
    total += (4 * Z + 5 * V) / 4.0 * n_s                       # w / x updates on the s-block, four iterations per pass (kWxDefer):
    #                                                            reads 4 u + w0 + w1 + x, writes w0 + w1 + x per 4 iterations
    # V-cycle on the Schur block, level by level until the first level handled by the LDS tail (<= ~6k rows) / last level
    for lv in range(len(L)):
        ns_l = L[lv].n_s
        out = Z if lv == 0 else V                              # the cycle's result is the s-block of z
        if ns_l <= 6000 or lv == len(L) - 1:
            total += (V + out) * ns_l                          # tail: r in, x out (matrices of the tail levels stay in L2)
            break
        B = L[lv].B.tocsr()
        nnzS = ((abs(B) @ abs(B).T) + sp.identity(ns_l)).nnz   # pattern of aW + B diag(M)^-1 B^T
        nc = L[lv + 1].n_s
        # fp32 intermediates (k::vc_* kernels): the level's iterate x and residual live in fp32
        total += mat(nnzS, ns_l) + 8.0 * ns_l + (V + F) * ns_l                 # pre-smoothing (one pass): r in, x out
        total += mat(nnzS, ns_l) + (V + 2 * F) * ns_l + V * nc                 # residual (r, x in; res out) + fused restriction
        total += mat(nnzS, ns_l) + 2 * F * ns_l + V * nc                       # res - (S P) xc   (S P has the pattern of S)
        total += mat(nnzS, ns_l) + 12.0 * ns_l + (2 * F + V + out) * ns_l + V * nc   # post-smoothing + coarse correction + dot
    return total


def z_bytes():
    from parelagmc_amd import capi
    return int(capi.load_library().pmc_krylov_z_bytes())


def lib_sha256():
    import hashlib
    with open(os.path.join(ROOT, "parelagmc_amd", "lib", "libpmc.so"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def csrc_sha256():
    """hash of the sources libpmc.so is built from (a relinked library of the same sources is the same kernels)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "parelagmc_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "pmc.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


_TRAFFIC = {}


def traffic_entry(key):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json, written by scripts/collect_profiles.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs).  The file is stamped with the sha256 of the libpmc.so those
    passes ran: a different library -> the counters describe other kernels -> traffic is null (and a warning), never a stale
    number."""
    if not _TRAFFIC:
        try:
            _TRAFFIC.update(json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))))
        except Exception:   # noqa: BLE001
            _TRAFFIC["error"] = True
        stamp = _TRAFFIC.get("libpmc_sha256")
        _TRAFFIC["_ok"] = bool(stamp) and (stamp == lib_sha256() or _TRAFFIC.get("csrc_sha256") == csrc_sha256())
        if not _TRAFFIC["_ok"]:
            print("bench: profiles/pmc_traffic.json was not collected with this libpmc.so (stamp "
                  f"{str(stamp)[:12]}, library {lib_sha256()[:12]}): roofline.traffic = null; run scripts/make_profiles.sh + "
                  "scripts/collect_profiles.py", file=sys.stderr)
    if not _TRAFFIC.get("_ok"):
        return None
    return _TRAFFIC.get(key, {}).get("hbm_bytes_per_launch")


def traffic_provenance():
    traffic_entry("")
    return {"file": "profiles/pmc_traffic.json", "matches_running_library": bool(_TRAFFIC.get("_ok")),
            "libpmc_sha256": _TRAFFIC.get("libpmc_sha256"), "csrc_sha256": _TRAFFIC.get("csrc_sha256"),
            "head": _TRAFFIC.get("head")}


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
    # The headline is the RAW event bracket (event record, launch, event record): a measurement, and an upper bound of the
    # kernel's duration.  What an event pair costs on that stream is measured by the empty bracket behind every timed launch;
    # the bracket net of it is an ESTIMATE (it over-corrects by ~2.5 % against the rocprofv3 average of the same launches,
    # profiles/rNN_bench_s1_kernel_stats.csv) and is reported beside the headline, never as it.
    raw_ms = solo_ms / solo_launches if solo_launches > 0 else k_ms
    gap = gap_ms / solo_launches if solo_launches > 0 else 0.0
    # the in-loop launches read their input - a preconditioned vector - in its storage width (fp32), the isolated ones fp64
    loop_bytes = k_bytes - nb * (8.0 - z_bytes()) * (L.n_u + L.n_s)
    ach = loop_bytes / (raw_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"pmc::sell_spmm_kernel<{nb}, 0, 0, true, 1, ...> (tag 1 = block operator K5 as launched "
                                     "by the MINRES loop: fused <u, Au>, diagonal-last, non-temporal streams by size; one lane "
                                     "alone on the GPU; profile rows with this prefix)",
           "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
           "traffic": traffic_entry(f"r{refine}_nb{nb}_inloop"), "traffic_provenance": traffic_provenance(),
           "bytes_per_launch": loop_bytes, "avg_kernel_ms": raw_ms, "launches": solo_launches,
           "timing": "raw HIP-event bracket around every in-loop launch",
           "event_overhead_ms": gap, "frac_net_of_event_overhead": loop_bytes / (max(raw_ms - gap, 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
           "isolated": {"kernel": f"pmc::sell_spmm_kernel<{nb}, 0, 0, false, 2, ...> launched back to back",
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


def darcy_operator_roofline(ctx, smp, ds, level=0, nb=16):
    """Roofline block of the Darcy operator (solver->Mult, src/DarcySolver.cpp:479,629-631) as the MINRES loop of SolveFwd
    launches it: the u-rows y_u = M(k) x_u + B^T x_p with the fused <x, Ax> (eg_pair_spmm_kernel - M(k) element-grouped, never
    materialised), every in-loop launch of two solo batches of nb realizations bracketed by HIP events on the solve's stream
    (one lane alone on the GPU).  Algorithmic bytes: pmc_darcy_operator_bytes."""
    xi = ctx.empty(nb * smp.xi_size(level))
    s = ctx.empty(nb * smp.SampleSize(level))
    smp.Sample(level, first_id=900000, nbatch=nb, out=xi)
    smp.Eval(level, xi, xi_level=level, s_out=s)
    ds.SolveFwd(level, s, nbatch=nb)                   # warm-up at this width
    ds.set_operator_timing(True)
    ds.operator_time()
    its = []
    for _ in range(2):
        _, _, st = ds.SolveFwd(level, s, nbatch=nb, return_stats=True)
        check_stats(st, "the in-loop Darcy operator pass")
        its.append(max(t[0] for t in st))
    ms, n, gap = ds.operator_time()
    ds.set_operator_timing(False)
    nbytes = ds.operator_bytes(level, nb)
    raw = ms / max(n, 1)
    ach = nbytes / (raw * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": f"pmc::eg_pair_spmm_kernel<{nb}, true, ...> (u-rows [M(k) | B^T] x of the Darcy operator as "
                                      f"launched by the MINRES loop on level {level}: element-grouped M(k), fused <x, Ax>; one lane "
                                      "alone on the GPU, p-rows B x_u behind it on the same stream while timed)",
            "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
            "traffic": traffic_entry(f"c3_eg_nb{nb}_inloop"), "traffic_provenance": traffic_provenance(),
            "bytes_per_launch": nbytes, "avg_kernel_ms": raw, "launches": n, "timing": "raw HIP-event bracket around every in-loop launch",
            "event_overhead_ms": gap / max(n, 1),
            "frac_net_of_event_overhead": nbytes / (max(raw - gap / max(n, 1), 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
            "minres_iterations": its}


def mlmc_config3(seed, lanes=4, opts=None, farm=None, batch=256, roofline=False):
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
    ns = [64 * world, 256 * world, 1024 * world]
    mgr.InitRun(ns)                                  # warm-up: allocations at the widths of the timed round
    mgr.Reset()
    ph0 = [mgr.phase_times(l) for l in range(3)]
    l0 = ctxs[0].lib.pmc_kernel_launches()
    t0 = time.perf_counter()
    r = mgr.InitRun(ns)
    dt = time.perf_counter() - t0
    launches = int(ctxs[0].lib.pmc_kernel_launches() - l0)
    ph1 = [mgr.phase_times(l) for l in range(3)]
    widths = [min(sm[0].BatchWidth(l), dr[0].BatchWidth(l)) for l in range(3)]   # what the manager hands over: the smaller
    out = {"workload": "MLMC Darcy + SPDE sampler, cube_hex 64^3/32^3/16^3 (1 060 864 / 134 144 / 17 152 DoF), lognormal, "
                       f"eff_perm QoI, InitRun {ns}, {lanes} lanes, realizations per launch per level {widths}"
                       + (f", sharded over {world} ranks" if farm else ""),
           "realizations_per_s": sum(ns) / dt, "seconds": dt, "estimate": r["estimate"],
           "nsamples_after_allreduce": [int(x) for x in r["nsamples"]],
           "realizations_per_launch_per_level": widths,
           "kernel_launches_per_round": launches, "kernel_launches_per_s": launches / dt,
           "seconds_per_sample_per_level": [float(x) for x in r["cost"]], "varY": [float(x) for x in r["varY"]],
           # the reference's per-level TimeManager entries (src/PDESampler.cpp:328-333, src/DarcySolver.cpp:231-243): device ms
           # of the timed round, summed over the lanes of this rank (HIP events, pmc_stats.solve_ms / setup_ms)
           "phase_timers_ms": [{k: ph1[l][k] - ph0[l][k] for k in ph1[l]} for l in range(3)]}
    if roofline:
        try:
            out["roofline"] = darcy_operator_roofline(ctxs[0], sm[0], dr[0], 0, 16)
        except Exception as e:   # noqa: BLE001
            out["roofline"] = {"error": repr(e)}
    mgr.close()
    for x in dr + sm:
        x.close()
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


def level_rates(lanes, levels, nrep, darcy):
    """The reference's own timing harness shape (examples/SPE10/SPE10_PDESampler_Performance.cpp:163-175: per level,
    nsamples x (Sample + Eval)), with `lanes` = [(ctx, sampler, darcy solver or None)] working concurrently: every lane
    draws and evaluates `nrep` batches of the level's launch width (+ SolveFwd when darcy).  Device-resident buffers."""
    out = []
    for lvl in levels:
        w = lanes[0][1].BatchWidth(lvl)
        bufs = [(c.empty(w * sm.xi_size(lvl)), c.empty(w * sm.SampleSize(lvl))) for c, sm, _ in lanes]
        res = [None] * len(lanes)

        def work(i, reps, first):
            c, sm, ds = lanes[i]
            xi, sf = bufs[i]
            st_all, sq_all = [], []
            for r in range(reps):
                sm.Sample(lvl, first_id=first + (r * len(lanes) + i) * w, nbatch=w, out=xi)
                st_all += sm.Eval(lvl, xi, xi_level=lvl, s_out=sf, return_stats=True)[1]
                if ds is not None:
                    sq_all += ds.SolveFwd(lvl, sf, nbatch=w, return_stats=True)[2]
            c.synchronize()
            res[i] = (st_all, sq_all)

        def run(reps, first):
            th = [threading.Thread(target=work, args=(i, reps, first)) for i in range(len(lanes))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        run(1, 0)                                   # warm-up: allocations at this width
        t0 = time.perf_counter()
        run(nrep, 10 ** 6)
        dt = time.perf_counter() - t0
        st = [t for r in res for t in r[0]]
        sq = [t for r in res for t in r[1]]
        n = nrep * w * len(lanes)
        e = {"level": lvl, "realizations": n, "realizations_per_launch": w, "realizations_per_s": n / dt,
             "sampler_iterations_mean": float(np.mean([t[0] for t in st])), "sampler_all_converged": all(t[1] == 1 for t in st)}
        if darcy:
            e.update({"darcy_iterations_mean": float(np.mean([t[0] for t in sq])), "darcy_all_converged": all(t[1] == 1 for t in sq)})
        out.append(e)
    return out


def cpu_level_rates(sp, dp, seed, levels, per_core):
    """CPU column of a multi-level configuration: the C restatement of the reference's solver (oracle/c/pmc_ref.c) on a
    bounded sample per level, farmed over the host cores - sampler solve, plus (dp given) per-sample M(k), elimination,
    preconditioner rebuild and Darcy solve as src/DarcySolver.cpp:472-649 does."""
    from oracle.cport import CPort, DarcyCPort
    from oracle.rng_oracle import normal_fill
    cores = host_cores()
    cs = CPort(sp)
    cd = DarcyCPort(dp) if dp is not None else None
    out = []
    for lvl, pc in zip(levels, per_core):
        ns = pc * cores
        n = sp.levels[lvl].n_s
        xi = np.stack([normal_fill(n, seed, i, lvl) for i in range(ns)])
        t0 = time.perf_counter()
        sol, it = cs.solve(lvl, cs.rhs(lvl, lvl, xi), nthreads=cores)
        e = {"level": lvl, "realizations": ns, "sampler_iterations_mean": float(np.mean(np.abs(it))),
             "sampler_all_converged": bool(np.all(it > 0))}
        if cd is not None:
            field = sol[:, sp.levels[lvl].n_u:]
            if dp.levels[lvl].n_p != field.shape[1]:     # projected samplers: the CPU column times the solves, k == exp(0)
                field = np.zeros((ns, dp.levels[lvl].n_p))
            _, itd = cd.solve(lvl, np.exp(field), nthreads=cores)
            e.update({"darcy_iterations_mean": float(np.mean(np.abs(itd))), "darcy_all_converged": bool(np.all(itd > 0))})
        dt = time.perf_counter() - t0
        e.update({"seconds": dt, "realizations_per_s": ns / dt})
        out.append(e)
    return {"kind": "port", "cores": cores, "levels": out,
            "sample": f"{[pc * cores for pc in per_core]} realizations on levels {list(levels)} (MINRES 300/1e-6 + BJ[symGS x3 | "
                      "V-cycle over the caller's levels]; iteration counts negative = cap reached); the reference's BoomerAMG "
                      "is not available to the C restatement, on stretched cells its geometric V-cycle needs more iterations"}


def config4(seed, lanes=4, cpu=True):
    """BASELINE config 4 at full size on ONE GPU: EmbeddedPDESampler on cube_tet_embed refined 4 x (831 488 tets, 2.5 M DoF
    on the finest level), 3 Monte Carlo levels, lognormal."""
    from parelagmc_amd import capi
    from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json
    h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), 4)
    sp = build_sampler_problem(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=3)
    L = []
    for _ in range(lanes):
        c = capi.Context(0, seed=seed)
        L.append((c, capi.PDESampler(c, sp, projection="gather"), None))
    out = {"workload": "EmbeddedPDESampler cube_tet_embed r=4, DoF " + str([lv.n_u + lv.n_s for lv in sp.levels[:3]]) +
                       f", original elements {[len(i) for i in sp.orig_index[:3]]}, per level nsamples x (Sample + Eval), {lanes} lanes",
           "levels": level_rates(L, (0, 1, 2), nrep=4, darcy=False)}
    for c, sm, _ in L:
        sm.close()
        c.close()
    if cpu:
        try:
            out["cpu_baseline"] = cpu_level_rates(sp, None, seed, (0, 1, 2), (1, 2, 8))
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def config5(seed, lanes=4, cpu=True):
    """BASELINE config 5 at full size on ONE GPU: SPE10-shaped box 1200 x 2200 x 170, 7 x 27 x 10 coarse cells refined 3 x
    (56 x 216 x 80 = 967 680 elements, 3.9 M Darcy DoF), L2ProjectionPDESampler on the box enlarged by one coarse cell per side,
    correlation length 100, 4 levels, k_ref == 1 (spe_perm.dat is not shipped)."""
    from parelagmc_amd import capi, host_api
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem, l2_projection_hierarchy
    nx, ny, nz = 7, 27, 10
    hx, hy, hz = 1200.0 / nx, 2200.0 / ny, 170.0 / nz
    ho = build_hierarchy(box_mesh([nx, ny, nz], [1200.0, 2200.0, 170.0], "hex"), 3)
    he = build_hierarchy(box_mesh([nx + 2, ny + 2, nz + 2], [1200.0 + 2 * hx, 2200.0 + 2 * hy, 170.0 + 2 * hz], "hex",
                                  origin=[-hx, -hy, -hz]), 3)
    sp = build_sampler_problem(he, corlen=100.0, lognormal=True)
    ops = l2_projection_hierarchy(ho, he)
    dp = build_darcy_problem(ho, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0])
    L = []
    for _ in range(lanes):
        c = capi.Context(0, seed=seed)
        L.append((c, capi.PDESampler(c, sp, projection="l2", l2_ops=ops), capi.DarcySolver(c, dp)))
    out = {"workload": "SPE10-shaped box 56 x 216 x 80, L2ProjectionPDESampler (sampler DoF " + str([lv.n_u + lv.n_s for lv in sp.levels]) +
                       ") + DarcySolver (DoF " + str([lv.ndofs for lv in dp.levels]) + f"), 4 levels, algebraic Schur hierarchies "
                       f"on the stretched cells, per level nsamples x (Sample + Eval + SolveFwd), {lanes} lanes",
           "levels": level_rates(L, (0, 1, 2, 3), nrep=2, darcy=True)}
    mgr = host_api.MLMCManager(4, sampler=L[0][1], solver=L[0][2], wall_time=True)
    for c, sm, ds in L[1:]:
        mgr.add_lane(sm, ds)
    ns = [16, 32, 64, 128]
    mgr.InitRun(ns)
    mgr.Reset()
    t0 = time.perf_counter()
    r = mgr.InitRun(ns)
    dt = time.perf_counter() - t0
    out["mlmc_round"] = {"nsamples": ns, "realizations_per_s": sum(ns) / dt, "seconds": dt, "estimate": r["estimate"],
                         "seconds_per_sample_per_level": [float(x) for x in r["cost"]], "timers": mgr.PrintTimers()}
    mgr.close()
    for c, sm, ds in L:
        ds.close()
        sm.close()
        c.close()
    if cpu:
        try:
            out["cpu_baseline"] = cpu_level_rates(sp, dp, seed, (0, 1, 2, 3), (1, 1, 2, 8))
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


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
    ap.add_argument("--batch", type=int, default=32,
                    help="realizations per plugin call and lane (sampler levels up to 5 M rows run 32 per launch)")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent batches in flight per GPU (one context + host thread each): the launch-latency-"
                         "bound coarse-level kernels of one batch overlap the bandwidth-bound kernels of the other")
    ap.add_argument("--refine", type=int, default=5, help="uniform refinements of cube_tet (5 -> 595 968 DoF)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mlmc", action="store_true", help="skip the secondary config-3 (MLMC Darcy+SPDE) figure")
    ap.add_argument("--no-r6", action="store_true", help="skip the HBM-bound point (cube_tet r=6, 4.74 M DoF) under extra.r6")
    ap.add_argument("--all-configs", action="store_true",
                    help="also run BASELINE configs 4 (EmbeddedPDESampler, 2.5 M DoF) and 5 (SPE10-shaped Darcy MLMC, 3.9 M DoF) "
                         "at full size on this GPU with a bounded CPU sample beside each: extra.c4 / extra.c5 (minutes)")
    ap.add_argument("--only-config", type=int, choices=(4, 5), default=None,
                    help="with --all-configs: run only this one of the two (each takes minutes; a GPU call has a time limit)")
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
                m, probs = mlmc_config3(args.seed, roofline=True)
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
    if rank == 0 and world == 1 and args.all_configs:
        for name, fn in (("c4", config4), ("c5", config5)):
            if args.only_config is not None and name != f"c{args.only_config}":
                continue
            try:
                extra[name] = fn(args.seed, cpu=not args.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001
                extra[name] = {"error": repr(e)}
    if rank == 0 and extra:
        out["extra"] = extra
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
