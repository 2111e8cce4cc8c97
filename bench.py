#!/usr/bin/env python3
"""Headline benchmark: MC samples/sec of the SPDE Matérn sampler (BASELINE.json config 2:
PDESampler on cube_tet refined 5x -> 595 968 DoF, single Monte Carlo level) on N MI355X.

A step = Sample + Eval of `--streams` batches of `--batch` realizations, white noise drawn on the device so
all inputs are resident in HBM when the timed region starts.  Contract: W untimed warm-up steps,
then exactly K steps bracketed by barrier + torch.cuda.synchronize(); MAX over ranks; rank 0
prints ONE JSON line.  The timed region carries no instrumentation; the roofline figures are taken in
separate short passes after it.  Behind the headline the default run (N = 1) also measures, under "extra", every other
BASELINE configuration on this GPU with bounded work: config 3 (MLMC Darcy + SPDE), the 1 060 864-DoF hex sampler point,
the HBM-bound r = 6 point, configs 4 and 5 (GPU legs), the all-fp64 storage option and the drop-in one-realization-per-call
path; the finite-element SETUP of those configurations (numpy, no GPU) is built by background worker processes while the
GPU works on the earlier ones.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL),
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
WX_DEFER = 8.0       # csrc/kernels.hpp::kWxDefer: MINRES iterations whose w / x updates one pass applies


LINE_CAP = 4096     # the driver keeps ~8 000 characters of stdout: the ONE JSON line stays well below that (BENCH_r04: a 21.7 kB
#                     line lost its head - metric, value, roofline - and was recorded as unparsed)


def _r(x, nd=4):
    """a scalar for the compact line: floats rounded to nd significant digits, everything else as it is"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{nd}g}")


def _get(d, *path):
    for k in path:
        if isinstance(d, dict):
            d = d.get(k)
        elif isinstance(d, list) and isinstance(k, int) and -len(d) <= k < len(d):
            d = d[k]
        else:
            return None
    return d


def compact_line(full):
    """The ONE stdout line of bench.py: headline, config, roofline, cpu_baseline and an `extra` of SCALARS (and short lists of
    scalars) only, at most LINE_CAP bytes whatever the rank count.  Everything else - per-rank dicts, timer tables, prose, the
    nested roofline blocks of every secondary figure - is in the full record (bench_full.json, path in `full_record`)."""
    cfg = full.get("config", {})
    rf = full.get("roofline", {}) or {}
    hybrid = cfg.get("solver") == "hybridization"
    ex = full.get("extra", {}) or {}
    sad = ex.get("saddle_point_minres", {}) if hybrid else {}
    # K5 on the saddle-point operator A (the CSR SpMM BASELINE.json's metric names): the headline's own block when the
    # headline runs the default solver, extra.saddle_point_minres otherwise
    saddle_k5 = _get(sad, "roofline", "frac") if hybrid else rf.get("frac")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 6)
    out["config"] = {"workload": str(cfg.get("workload", ""))[:200], "solver": cfg.get("solver"), "batch": cfg.get("batch"),
                     "streams": cfg.get("streams"), "precond_storage": cfg.get("precond_storage"),
                     "mean_minres_iterations": _r(cfg.get("mean_minres_iterations")), "parallelism": cfg.get("parallelism"),
                     "tolerance": cfg.get("tolerance")}
    out["roofline"] = {"bound": "hbm", "kernel": str(rf.get("kernel", ""))[:80], "achieved": _r(rf.get("achieved")),
                       "peak": rf.get("peak", PEAK_GBS), "unit": "GB/s", "frac": _r(rf.get("frac")),
                       "traffic": _r(rf.get("traffic"), 6), "bytes_per_launch": _r(rf.get("bytes_per_launch"), 9),
                       "avg_kernel_ms": _r(rf.get("avg_kernel_ms")), "launches": rf.get("launches"),
                       "event_overhead_ms": _r(rf.get("event_overhead_ms")),
                       "frac_net_of_event_overhead": _r(rf.get("frac_net_of_event_overhead")),
                       "operator_frac": _r(_get(rf, "operator", "frac")) if hybrid else _r(rf.get("frac")),
                       "operator_traffic": _r(_get(rf, "operator", "traffic"), 6) if hybrid else _r(rf.get("traffic"), 6),
                       "operator_bytes_per_launch": _r(_get(rf, "operator", "bytes_per_launch"), 9) if hybrid else None,
                       "saddle_k5_frac": _r(saddle_k5), "spmv_nb1_frac": _r(_get(sad if hybrid else full, "roofline", "spmv_nb1", "frac")),
                       "solver_frac": _r(_get(rf, "solver", "frac")),
                       "traffic_matches_library": _get(rf, "traffic_provenance", "matches_running_library")}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit", "samples/s"), "cores": cb.get("cores"),
                               "kind": cb.get("kind"), "solver": str(cb.get("solver", ""))[:72], "iterations": _r(cb.get("iterations")),
                               "sample": str(cb.get("sample", ""))[:160]}
        for k in ("saddle_value", "saddle_iterations", "error"):
            if cb.get(k) is not None:
                out["cpu_baseline"][k] = _r(cb[k]) if k != "error" else str(cb[k])[:120]
    if full.get("n_gpus", 1) > 1:
        sec = [r.get("seconds") for r in full.get("ranks", []) if isinstance(r, dict)]
        out["rank_seconds"] = {"min": _r(min(sec)), "max": _r(max(sec))} if sec else None
        out["devices"] = full.get("devices")
    e = {}

    def put(key, v):
        if v is not None:
            e[key] = v

    def lv(block, key):
        L = _get(block, "levels")
        return [_r(x.get(key)) for x in L] if isinstance(L, list) else None
    put("saddle_value", _r(sad.get("value")))
    put("saddle_iterations", _r(sad.get("mean_minres_iterations")))
    put("saddle_solver_frac", _r(_get(sad, "roofline", "solver", "frac")))
    put("hybrid_value", _r(_get(ex, "hybridization", "value")))
    put("fp64_value", _r(_get(ex, "fp64_storage", "value")))
    put("dropin_eval_ms", _r(_get(ex, "dropin_nb1", "config2", "ms_per_Eval")))
    put("dropin_samples_per_s", _r(_get(ex, "dropin_nb1", "config2", "samples_per_s")))
    put("dropin_saddle_eval_ms", _r(_get(ex, "dropin_nb1", "config2_saddle_point_minres", "ms_per_Eval")))
    c3 = ex.get("mlmc_config3", {})
    put("c3_value", _r(c3.get("realizations_per_s")))
    put("c3_cpu", _r(_get(c3, "cpu_baseline", "realizations_per_s")))
    put("c3_eg_spmm_frac", _r(_get(c3, "roofline", "frac")))
    put("c3_eg_poly_frac", _r(_get(c3, "roofline", "m_block_polynomial", "frac")))
    put("c3_eg_poly_traffic", _r(_get(c3, "roofline", "m_block_polynomial", "traffic"), 6))
    put("c3_dropin_round", _r(_get(ex, "dropin_nb1_config3", "round_64_256_1024_realizations_per_s")))
    dh = ex.get("darcy_hybridization", {})
    put("c3_darcy_saddle_ms", _r(_get(dh, "saddle_point", "ms_per_realization")))
    put("c3_darcy_saddle_iterations", _r(_get(dh, "saddle_point", "iterations_mean")))
    put("c3_darcy_hybrid_ms", _r(_get(dh, "hybridized", "ms_per_realization")))
    put("c3_darcy_hybrid_iterations", _r(_get(dh, "hybridized", "iterations_mean")))
    hx = ex.get("hex64", {})
    put("hex64_value", _r(hx.get("value")))
    put("hex64_k5_frac", _r(_get(hx, "roofline", "frac")))
    put("hex64_cpu", _r(_get(hx, "cpu_baseline", "value")))
    put("c4_values", lv(ex.get("c4"), "realizations_per_s"))
    put("c4_iterations", lv(ex.get("c4"), "sampler_iterations_mean"))
    put("c4_other_solver_values", lv(_get(ex, "c4", "other_solver"), "realizations_per_s"))
    put("c5_values", lv(ex.get("c5"), "realizations_per_s"))
    put("c5_mlmc_round", _r(_get(ex, "c5", "mlmc_round", "realizations_per_s")))
    r6 = ex.get("r6", {})
    put("r6_value", _r(r6.get("value")))
    put("r6_iterations", _r(r6.get("mean_minres_iterations")))
    put("r6_frac", _r(_get(r6, "roofline", "frac")))
    put("r6_operator_frac", _r(_get(r6, "roofline", "operator", "frac")))
    put("r6_solver_frac", _r(_get(r6, "roofline", "solver", "frac")))
    put("r6_cpu", _r(_get(r6, "cpu_baseline", "value")))
    put("r6_cpu_saddle", _r(_get(r6, "cpu_baseline", "saddle_value")))
    put("r6_other_solver_value", _r(_get(r6, "other_solver", "value")))
    put("r6_saddle_k5_frac", _r(_get(r6, "other_solver", "roofline", "frac")))
    fm = ex.get("mlmc_farm", {})
    put("farm_value", _r(fm.get("realizations_per_s")))
    put("farm_collective", str(fm["collective"])[:60] if fm.get("collective") else None)
    put("farm_allreduces_in_round", fm.get("allreduces_in_round"))
    put("farm_allreduce_ms_max", _r(max(fm["allreduce_ms_per_rank"])) if fm.get("allreduce_ms_per_rank") else None)
    put("farm_error", str(fm["error"])[:120] if fm.get("error") else None)
    errs = sorted(k for k, v in ex.items() if isinstance(v, dict) and "error" in v)
    put("errors", errs or None)
    if e:
        out["extra"] = e
    out["full_record"] = full.get("full_record")
    line = json.dumps(out, separators=(",", ":"))
    while len(line) >= LINE_CAP and out.get("extra"):       # cannot happen with the fields above; the cap holds regardless
        out["extra"].pop(next(reversed(out["extra"])))
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < LINE_CAP, len(line)
    return line


def write_full_record(full):
    """the complete record (every nested block the compact line leaves out) -> bench_full.json beside bench.py and, when the
    directory exists, gpurun_out/bench_full.json; returns the path written first (None if neither is writable)"""
    first = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_full.json"), "w") as f:
                json.dump(full, f, indent=1)
            first = first or os.path.relpath(os.path.join(d, "bench_full.json"), ROOT)
        except OSError:
            pass
    return first


def build_problem(nref, extra_coarse=True):
    from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json
    mesh = mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json"))
    h = build_hierarchy(mesh, nref)
    # one Monte Carlo level (config 2); the coarser refinement levels only deepen the V-cycle
    return build_sampler_problem(h, corlen=0.1, lognormal=False, n_mc_levels=1)


def build_hybrid_problem(nref):
    """the hybridized form of build_problem's system (the reference's "Hybridization" solver option): level 0 only - the
    multiplier system brings its own algebraic hierarchy"""
    from parelagmc_amd.capi import library_hybrid_builder
    from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, mesh_from_json
    mesh = mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json"))
    # H, G, z from the library's own element-local elimination (pmc_hybrid_build: what a C++ caller of libpmc.so uses;
    # host code - loading the library does not open the GPU, scripts/r5/kfd_probe.py)
    return build_hybrid_sampler_problem(build_hierarchy(mesh, nref), corlen=0.1, lognormal=False, n_mc_levels=1,
                                        builder=library_hybrid_builder)


def is_hybrid(problem):
    return hasattr(problem.levels[0], "n_lambda")


def build_config3():
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
    sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
    return sp, dp


def build_config4():
    """(saddle-point form, hybridized form) of the same three levels"""
    from parelagmc_amd.capi import library_hybrid_builder
    from parelagmc_amd.fe import build_hierarchy, build_hybrid_sampler_problem, build_sampler_problem, mesh_from_json
    h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), 4)
    kw = dict(corlen=0.1, embedded=True, lognormal=True, n_mc_levels=3)
    return build_sampler_problem(h, **kw), build_hybrid_sampler_problem(h, builder=library_hybrid_builder, **kw)


def build_config5():
    from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem, l2_projection_hierarchy
    nx, ny, nz = 7, 27, 10
    hx, hy, hz = 1200.0 / nx, 2200.0 / ny, 170.0 / nz
    ho = build_hierarchy(box_mesh([nx, ny, nz], [1200.0, 2200.0, 170.0], "hex"), 3)
    he = build_hierarchy(box_mesh([nx + 2, ny + 2, nz + 2], [1200.0 + 2 * hx, 2200.0 + 2 * hy, 170.0 + 2 * hz], "hex",
                                  origin=[-hx, -hy, -hz]), 3)
    sp = build_sampler_problem(he, corlen=100.0, lognormal=True)
    ops = l2_projection_hierarchy(ho, he)
    dp = build_darcy_problem(ho, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0])
    return sp, ops, dp


def _setup_worker(kind):
    """runs in a worker PROCESS (numpy / scipy only, never touches the GPU): the one-time finite-element setup of a
    configuration - what the reference does once in BuildHierarchy (/root/reference/src/PDESampler.cpp:177-334,
    src/DarcySolver.cpp:60-244), outside every timed region"""
    t0 = time.perf_counter()
    out = {"r6": lambda: (build_problem(6), build_hybrid_problem(6)), "c3": build_config3, "c4": build_config4,
           "c5": build_config5}[kind]()
    return out, time.perf_counter() - t0


def _worker_init(cpus):
    try:
        if cpus:
            os.sched_setaffinity(0, cpus)
        os.nice(5)
    except Exception:   # noqa: BLE001
        pass


class SetupPool:
    """Background builders of the configurations' operators.  Started BEFORE this process touches the GPU (fresh `spawn`
    children that import numpy / scipy only), kept off the first CPUs so that the lanes' launch threads are not disturbed."""

    def __init__(self, kinds):
        import concurrent.futures as cf
        import multiprocessing as mp
        self.fut, self.seconds = {}, {}
        self.pool = None
        if not kinds:
            return
        allowed = sorted(os.sched_getaffinity(0))
        cpus = allowed[6:] if len(allowed) >= 12 else None
        self.pool = cf.ProcessPoolExecutor(max_workers=len(kinds), mp_context=mp.get_context("spawn"),
                                           initializer=_worker_init, initargs=(cpus,))
        for k in kinds:
            self.fut[k] = self.pool.submit(_setup_worker, k)

    def get(self, kind):
        """the configuration's problem objects (blocks until the worker has them; built inline without a pool)"""
        if kind not in self.fut:
            out, dt = _setup_worker(kind)
        else:
            try:
                out, dt = self.fut.pop(kind).result()
            except Exception as e:   # noqa: BLE001 - a lost worker must not cost the figure: build it here
                print(f"bench: setup worker for {kind} failed ({e!r}); building inline", file=sys.stderr)
                out, dt = _setup_worker(kind)
        self.seconds[kind] = dt
        return out

    def wait(self):
        """block until every worker has delivered: the results arrive in THIS process through a pipe and are unpickled by
        the executor's thread under the GIL - hundreds of MB for the large configurations, which would otherwise land in
        the timed region and stall the lanes' launch threads (measured: 2 440 instead of 2 700 samples/s)"""
        if self.fut:
            import concurrent.futures as cf
            cf.wait(list(self.fut.values()))

    def close(self):
        if self.pool:
            self.pool.shutdown(wait=False, cancel_futures=True)
            self.pool = None


def gpu_numa_cpus(local_rank):
    """CPUs of the NUMA node the rank's GPU hangs off (KFD topology -> DRM render node -> numa_node), restricted to the
    CPUs this process may use; None when the box does not say.  Read from sysfs only: usable before any HIP call."""
    top = "/sys/class/kfd/kfd/topology/nodes"
    gpus = []
    for node in sorted(os.listdir(top), key=int):
        props = dict(line.split()[:2] for line in open(os.path.join(top, node, "properties")) if len(line.split()) >= 2)
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(int(props["drm_render_minor"]))
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    order = [int(x) for x in vis.split(",")] if vis and all(x.strip().isdigit() for x in vis.split(",")) else list(range(len(gpus)))
    minor = gpus[order[local_rank % len(order)]]
    numa = int(open(f"/sys/class/drm/renderD{minor}/device/numa_node").read())
    if numa < 0:
        return None
    cpus = set()
    for part in open(f"/sys/devices/system/node/node{numa}/cpulist").read().strip().split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= os.sched_getaffinity(0)
    # a rank needs room for its lane threads and its setup worker: with fewer than 8 usable CPUs on that node stay unpinned
    return {"numa_node": numa, "cpus": sorted(cpus)} if len(cpus) >= 8 else None


def pin_rank_to_gpu_numa(local_rank):
    """Called by a rank BEFORE it imports torch or creates a pmc_ctx: 8 ranks x 4 lane threads issue ~57 k launches/s each,
    and a launch thread on the other socket pays the inter-socket hop on every doorbell.  Returns what was done (for the
    rank's record in the JSON line); never fatal."""
    try:
        info = gpu_numa_cpus(local_rank)
        if info is None:
            return {"pinned": False, "why": "no NUMA information for the GPU"}
        os.sched_setaffinity(0, info["cpus"])
        return {"pinned": True, "numa_node": info["numa_node"], "ncpus": len(info["cpus"])}
    except Exception as e:   # noqa: BLE001
        return {"pinned": False, "why": repr(e)}


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


def cpu_baseline(problem, seed, nsamples_per_core=12, hybrid_problem=None):
    """The reference's algorithms restated in C (oracle/c/pmc_ref.c), farmed over the host cores on a bounded sample of the
    same workload.  problem: saddle-point form -> the reference's default MINRES-BJ-GS; hybrid_problem (when the GPU figure
    beside it runs the hybridized solver): the reference's "Hybridization" entry, PCG + AMG V-cycle on H lambda = G f with
    back-substitution.  `value` is ALWAYS the solver the GPU figure ran (like for like, named in `solver`); the other one is
    reported as saddle_value."""
    from oracle.cport import CPort, HybridCPort
    from oracle.rng_oracle import normal_fill
    cores = host_cores()
    ns = nsamples_per_core * cores
    n = problem.levels[0].n_s
    xi = np.stack([normal_fill(n, seed, i, 0) for i in range(ns)])
    cp = CPort(problem)
    rhs = cp.rhs(0, 0, xi)
    if nsamples_per_core >= 4:
        cp.solve(0, rhs[:cores], nthreads=cores)      # warm-up (page in, thread pool); not for the one-per-core samples
    t0 = time.perf_counter()
    sol, iters = cp.solve(0, rhs, nthreads=cores)
    dt = time.perf_counter() - t0
    sad = {"value": ns / dt, "iterations": float(np.mean(np.abs(iters))), "seconds": dt,
           "solver": "MINRES(300,1e-6)+BJ[symGS x3 | V-cycle] on the saddle-point system"}
    out = {"unit": "samples/s", "cores": cores, "kind": "port"}
    if hybrid_problem is None:
        out.update({"value": sad["value"], "solver": sad["solver"], "iterations": sad["iterations"],
                    "sample": f"{ns} realizations ({nsamples_per_core} per core), {dt:.1f} s wall"})
        return out
    hc = HybridCPort(hybrid_problem)
    t0 = time.perf_counter()
    hc.hierarchy(0)
    setup = time.perf_counter() - t0
    f = hc.rhs(0, 0, xi)
    if nsamples_per_core >= 4:
        hc.solve(0, f[:cores], nthreads=cores)
    t0 = time.perf_counter()
    sh, ith = hc.solve(0, f, nthreads=cores)
    dth = time.perf_counter() - t0
    diff = float(np.linalg.norm(sh - sol[:, problem.levels[0].n_u:]) / np.linalg.norm(sh))
    out.update({"value": ns / dth, "solver": "PCG(300,1e-6)+AMG V(1,1) symGS on H lambda = G f, back-substitution "
                                             "(the reference's Hybridization entry; smoothed aggregation for BoomerAMG)",
                "iterations": float(np.mean(np.abs(ith))), "all_converged": bool(np.all(ith > 0) and np.all(iters > 0)),
                "operator_complexity": hc.operator_complexity(0), "amg_setup_seconds_once": setup,
                "saddle_value": sad["value"], "saddle_iterations": sad["iterations"], "saddle_solver": sad["solver"],
                "field_difference_between_the_two_cpu_solvers": diff,
                "sample": f"{ns} realizations ({nsamples_per_core} per core) with either solver: {dth:.1f} s / {dt:.1f} s wall"})
    return out


def solver_bytes_per_iteration(problem, nb, zb=4):
    """ALGORITHMIC bytes one MINRES iteration of a batch of nb realizations moves on level 0 (every
    operand once; matrices 12 B per stored nonzero + 4 B per row, vectors 8 nb B per row touched)."""
    import scipy.sparse as sp
    L = problem.levels
    V = 8.0 * nb
    Z = float(zb) * nb             # a preconditioned vector z: fp32 storage by default, fp64 with PMC_STORAGE_FP64
    F = float(zb) * nb             # an intermediate of the V-cycle (iterate, residual): the same storage option

    def mat(nnz, nrows):
        return 12.0 * nnz + 4.0 * nrows

    n_u, n_s = L[0].n_u, L[0].n_s
    n = n_u + n_s
    total = mat(L[0].nnz, n) + (Z + V) * n                    # K5: q = A u (u is a z; the dot takes u from the gathers)
    total += V * 4 * n                                         # v_new = c0 q + c1 v1 + c2 v0
    total += mat(L[0].M.nnz, n_u) + 8.0 * n_u + (V + Z) * n_u  # M-block: one-pass degree-2 polynomial, r in, z out
    total += (WX_DEFER * Z + 5 * V) / WX_DEFER * n_s           # w / x updates on the s-block, WX_DEFER iterations per pass (kWxDefer):
    #                                                            reads 8 u + w0 + w1 + x, writes w0 + w1 + x per 8 iterations
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


def hybrid_bytes_per_iteration(hp, smp, nb, zb=4):
    """ALGORITHMIC bytes one MINRES iteration on the multiplier system moves on the FINEST level of its V-cycle plus the
    Krylov part - a lower bound of the iteration's bytes (the coarser levels of the aggregation hierarchy, ~12 % of the finest
    level's rows, and the product with S P are left out), so the solver fraction derived from it is conservative."""
    L = hp.levels[0]
    n, nnz = L.n_lambda, L.H.nnz
    V, Z, F = 8.0 * nb, float(zb) * nb, float(zb) * nb
    post = smp.smoother_bytes(0, nb)
    nc = max(0.0, (post - 12.0 * nnz - 12.0 * n - (2 * F + V + Z) * n) / V)    # rows of the first coarse level
    # fp32 storage: the Lanczos update also writes an fp32 copy of the new vector, which the cycle's first two kernels read
    # instead of the fp64 one (csrc/solver.hpp: MinresWork::r32)
    R = F if zb == 4 else V
    total = 12.0 * nnz + 4.0 * n + (Z + V) * n                 # K5 on H: q = H u
    total += V * 4 * n + (F * n if zb == 4 else 0.0)           # v_new = c0 q + c1 v1 + c2 v0 (+ its fp32 copy)
    total += (WX_DEFER * Z + 6 * V) / WX_DEFER * n             # w / x updates on ALL rows, WX_DEFER iterations per pass
    total += 12.0 * nnz + 12.0 * n + (R + F) * n               # pre-smoothing: r in, x out
    total += 12.0 * nnz + 4.0 * n + (R + 2 * F) * n            # residual: r, x in; res out
    fused = bool(smp.vcycle_levels(0)[0]["fused_restriction"])
    # restriction P^T res: inside the residual kernel (multipliers renumbered, csrc/sparse.hip::agg_pack_rows: the segment
    # tables + the coarse rows written) or a separate product that reads the residual once more
    total += (8.0 * nc + V * nc) if fused else (12.0 * n + F * n + V * nc)
    total += 2 * F * n + V * nc                                # res - (S P) xc, without the bytes of S P
    total += post
    return total


def hybrid_roofline(farm, hp, nb, tag, next_batch, iters_total, dt):
    """Roofline block of the hybridized solver.  The largest single kernel of its iteration is the post-smoothing of the finest
    V-cycle level (k::vc_postsmooth32): `achieved` = its algorithmic bytes / the raw HIP-event bracket around every launch
    of three solo batches of lane 0.  `operator` = the Krylov operator K5 on H (the CSR SpMM the metric names) bracketed in
    the same batches; `solver` = the whole iteration."""
    ctx, smp = farm.lanes[0][0], farm.lanes[0][1]
    L = hp.levels[0]
    zb = smp.z_bytes()
    smp.set_operator_timing(True)
    smp.operator_time()
    smp.smoother_time()
    for j in range(3):
        check_stats(farm.one_batch(0, next_batch + j), "the in-loop kernel pass")
    gap_ms = smp.operator_event_overhead()
    op_ms, op_n = smp.operator_time()
    sm_ms, sm_n, sm_gap = smp.smoother_time()
    smp.set_operator_timing(False)
    x = ctx.array(np.random.default_rng(0).standard_normal(nb * L.n_lambda))
    _, k_ms, k_bytes = smp.Mult(0, x, repeat=50)
    op_bytes = k_bytes - nb * (8.0 - zb) * L.n_lambda
    sm_bytes = smp.smoother_bytes(0, nb)
    op_raw = op_ms / max(op_n, 1)
    sbytes = hybrid_bytes_per_iteration(hp, smp, nb, zb)
    kn, grp = min(nb, 32), (f", {nb // 32} column groups of 32 per launch" if nb > 32 else "")
    operator = {"kernel": f"pmc::sell_spmm_kernel<{kn}, 0, 0, true, 1, ...> = K5 on H (n_lambda = {L.n_lambda}, "
                          f"nnz = {L.H.nnz}) with the fused <u, Hu>, in the MINRES loop{grp}",
                "achieved": op_bytes / (op_raw * 1e-3) / 1e9, "frac": op_bytes / (op_raw * 1e-3) / 1e9 / PEAK_GBS,
                "bytes_per_launch": op_bytes, "avg_kernel_ms": op_raw, "launches": op_n,
                "event_overhead_ms": gap_ms / max(op_n, 1),
                "frac_net_of_event_overhead": op_bytes / (max(op_raw - gap_ms / max(op_n, 1), 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
                "traffic": traffic_entry(f"{tag}_hyb_k5_nb{nb}_inloop"),
                "isolated": {"achieved": k_bytes / (k_ms * 1e-3) / 1e9, "avg_kernel_ms": k_ms,
                             "frac": k_bytes / (k_ms * 1e-3) / 1e9 / PEAK_GBS}}
    solver = {"bytes_per_iteration": sbytes, "batch_iterations": iters_total,
              "what": "finest V-cycle level + Krylov vectors only (a lower bound of the bytes: conservative)",
              "achieved": sbytes * iters_total / dt / 1e9, "frac": sbytes * iters_total / dt / 1e9 / PEAK_GBS}
    if sm_n == 0:
        # everything stored fp64 (PMC_STORAGE_FP64): the V-cycle runs the generic fp64 kernels, which are not bracketed -
        # the block is the operator's then
        out = dict(operator)
        out.update({"bound": "hbm", "peak": PEAK_GBS, "unit": "GB/s", "solver": solver,
                    "timing": "raw HIP-event bracket around every in-loop launch"})
        return out
    sm_raw = sm_ms / sm_n
    ach = sm_bytes / (sm_raw * 1e-3) / 1e9
    return {"bound": "hbm",
            "kernel": f"pmc::vc_poly2_kernel<{kn}, float, {'float' if zb == 4 else 'double'}, float, true, true, 0, 1> = "
                      "post-smoothing of the finest level of the multiplier V-cycle (one-pass degree-2 polynomial of the "
                      f"residual + coarse correction + fused <r, z>), as launched inside the MINRES loop{grp}; one lane alone on the GPU",
            "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
            "traffic": traffic_entry(f"{tag}_hyb_post_nb{nb}_inloop"), "traffic_provenance": traffic_provenance(),
            "bytes_per_launch": sm_bytes, "avg_kernel_ms": sm_raw, "launches": sm_n,
            "timing": "raw HIP-event bracket around every in-loop launch", "event_overhead_ms": sm_gap / max(sm_n, 1),
            # `frac` stays the raw bracket (kernel + the two event records); net of the measured cost of an empty bracket
            # it is the figure the rocprofv3 average of the same kernel gives (profiles/r05_bench_s1_kernel_stats.csv)
            "frac_net_of_event_overhead": sm_bytes / (max(sm_raw - sm_gap / max(sm_n, 1), 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
            "operator": operator, "solver": solver}


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


def in_threads(fn, n):
    """fn(0) ... fn(n - 1) on n host threads (ctypes releases the GIL inside the library); re-raises the first failure"""
    err = [None] * n

    def run(i):
        try:
            fn(i)
        except BaseException as e:   # noqa: BLE001
            err[i] = e
    if n == 1:
        run(0)
    else:
        th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    for e in err:
        if e is not None:
            raise e


class SamplerFarm:
    """`streams` independent batches in flight on one GPU: one context + sampler + buffers each."""

    def __init__(self, problem, dev, seed, nb, streams, world=1, rank=0, opts=None):
        """nb: realizations per plugin call and lane; 0 / None = what the sampler prefers for level 0
        (pmc_sampler_batch_width: the width one launch of its kernels carries)"""
        from parelagmc_amd import capi
        self.ns, self.world, self.rank = streams, world, rank
        self.n = problem.levels[0].n_s
        self.lanes = [None] * streams
        handles = [None] * streams

        def make(i):    # the lanes' handles are set up side by side (host-side re-layout of the operators, one copy each)
            c = capi.Context(dev, seed=seed)
            c.seed(seed, nparts=world, mypart=rank)
            handles[i] = (c, capi.PDESampler(c, problem, opts))
        in_threads(make, streams)
        self.nb = nb = int(nb) if nb else handles[0][1].BatchWidth(0)
        for i, (c, sm) in enumerate(handles):
            self.lanes[i] = (c, sm, c.empty(nb * self.n), c.empty(nb * self.n))

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


def operator_roofline(farm, problem, nb, tag, next_batch, solver_bytes, iters_total, dt):
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
    # the in-loop launches read their input - a preconditioned vector - in its storage width (fp32 by default), the
    # isolated ones fp64
    loop_bytes = k_bytes - nb * (8.0 - smp.z_bytes()) * (L.n_u + L.n_s)
    ach = loop_bytes / (raw_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"pmc::sell_spmm_kernel<{min(nb, 32)}, 0, 0, true, 1, ...> x {max(1, nb // 32)} column group(s) (tag 1 = block operator K5 as launched "
                                     "by the MINRES loop: fused <u, Au>, diagonal-last, non-temporal streams by size; one lane "
                                     "alone on the GPU; profile rows with this prefix)",
           "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
           "traffic": traffic_entry(f"{tag}_nb{nb}_inloop"), "traffic_provenance": traffic_provenance(),
           "bytes_per_launch": loop_bytes, "avg_kernel_ms": raw_ms, "launches": solo_launches,
           "timing": "raw HIP-event bracket around every in-loop launch",
           "event_overhead_ms": gap, "frac_net_of_event_overhead": loop_bytes / (max(raw_ms - gap, 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
           "isolated": {"kernel": f"pmc::sell_spmm_kernel<{min(nb, 32)}, 0, 0, false, 2, ...> launched back to back",
                        "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "frac": k_bytes / (k_ms * 1e-3) / 1e9 / PEAK_GBS,
                        "avg_kernel_ms": k_ms, "traffic": traffic_entry(f"{tag}_nb{nb}")},
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
    pms, pn, pgap = ds.poly_time()
    ds.set_operator_timing(False)
    nbytes = ds.operator_bytes(level, nb)
    raw = ms / max(n, 1)
    ach = nbytes / (raw * 1e-3) / 1e9
    pbytes = ds.poly_bytes(level, nb)
    praw = pms / max(pn, 1)
    poly = {"kernel": f"pmc::eg_poly2_kernel<{nb}, true, ...> (M-block polynomial of the preconditioner, z_u = D^-1 (c0 r - c1 M(k) "
                      "D^-1 r), element-grouped M(k), per-realization l1 diagonal gathered beside r, fused <r, z>; timed on the "
                      "solve's main stream)",
            "achieved": pbytes / (praw * 1e-3) / 1e9 if pn else None, "peak": PEAK_GBS, "unit": "GB/s",
            "frac": pbytes / (praw * 1e-3) / 1e9 / PEAK_GBS if pn else None, "bytes_per_launch": pbytes, "avg_kernel_ms": praw,
            "launches": pn, "event_overhead_ms": pgap / max(pn, 1), "traffic": traffic_entry(f"c3_egpoly_nb{nb}_inloop")}
    return {"bound": "hbm", "m_block_polynomial": poly,
            "kernel": f"pmc::eg_pair_spmm_kernel<{nb}, true, ...> (u-rows [M(k) | B^T] x of the Darcy operator as "
                                      f"launched by the MINRES loop on level {level}: element-grouped M(k), fused <x, Ax>; one lane "
                                      "alone on the GPU, p-rows B x_u behind it on the same stream while timed)",
            "achieved": ach, "peak": PEAK_GBS, "unit": "GB/s", "frac": ach / PEAK_GBS,
            "traffic": traffic_entry(f"c3_eg_nb{nb}_inloop"), "traffic_provenance": traffic_provenance(),
            "bytes_per_launch": nbytes, "avg_kernel_ms": raw, "launches": n, "timing": "raw HIP-event bracket around every in-loop launch",
            "event_overhead_ms": gap / max(n, 1),
            "frac_net_of_event_overhead": nbytes / (max(raw - gap / max(n, 1), 1e-9) * 1e-3) / 1e9 / PEAK_GBS,
            "minres_iterations": its}


def mlmc_config3(seed, probs, lanes=4, opts=None, farm=None, batch=256, roofline=False, reduce=None):
    """Secondary figure (BASELINE config 3): MLMC_Manager::InitRun with the SPDE sampler + Darcy QoI on cube_hex
    64^3 / 32^3 / 16^3, fixed sample counts, `lanes` concurrent streams.  Reported under "extra", never as `value`.
    farm = (world, rank, comm_ctx, device): the realizations of every level are sharded over the ranks and the accumulators
    are all-reduced through the library's own RCCL communicator of comm_ctx (MLMC_Manager::SetFarm with reduce == NULL), or
    through `reduce` (a rehearsal over gloo on a box with fewer GPUs than ranks).  probs = (sampler problem, Darcy problem)."""
    from parelagmc_amd import capi, host_api
    sp, dp = probs
    dev = farm[3] if farm else 0
    # farm: the manager's primary context is the one that carries the RCCL communicator
    ctxs = ([farm[2]] if farm else []) + [capi.Context(dev, seed=seed) for _ in range(lanes - (1 if farm else 0))]
    sm, dr = [None] * lanes, [None] * lanes

    def make(i):
        sm[i] = capi.PDESampler(ctxs[i], sp, opts)
        dr[i] = capi.DarcySolver(ctxs[i], dp, opts)
    in_threads(make, lanes)
    # batch = upper limit of realizations per plugin call; per level the manager hands over what the plugins prefer
    # (pmc_sampler_batch_width: 16 at a time on the bandwidth-bound 64^3 level, 64 on 32^3, 256 - eight column groups of 32
    # in one launch - on 16^3), cut so that every lane gets a share
    mgr = host_api.MLMCManager(3, sampler=sm[0], solver=dr[0], wall_time=True, batch=batch)
    for i in range(1, lanes):
        mgr.add_lane(sm[i], dr[i])
    world = 1
    if farm:
        world, rank = farm[0], farm[1]
        mgr.set_farm(world, rank, reduce)            # reduce == None -> pmc_allreduce_sum_f64 (RCCL) of ctxs[0]
    ns = [64 * world, 256 * world, 1024 * world]
    mgr.InitRun(ns)                                  # warm-up: allocations at the widths of the timed round
    mgr.Reset()
    red0 = mgr.farm_times()
    ph0 = [mgr.phase_times(l) for l in range(3)]
    l0 = ctxs[0].lib.pmc_kernel_launches()
    t0 = time.perf_counter()
    r = mgr.InitRun(ns)
    dt = time.perf_counter() - t0
    launches = int(ctxs[0].lib.pmc_kernel_launches() - l0)
    ph1 = [mgr.phase_times(l) for l in range(3)]
    red1 = mgr.farm_times()
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
    if farm:
        # the farm's one collective (MLMC_Manager::InitRun, before computeNSamplesMSE): wall time of this rank inside it -
        # waiting for the slowest rank included - and how many there were in the timed round (must be ONE)
        out["allreduce_ms"] = red1[0] - red0[0]
        out["allreduces_in_round"] = int(red1[1] - red0[1])
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
    return out


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


def config4(seed, sp, lanes=4, cpu=True, nrep=4):
    """BASELINE config 4 at full size on ONE GPU: EmbeddedPDESampler on cube_tet_embed refined 4 x (831 488 tets, 2.5 M DoF
    on the finest level), 3 Monte Carlo levels, lognormal.  sp: build_config4()."""
    from parelagmc_amd import capi
    L = [None] * lanes

    def make(i):
        c = capi.Context(0, seed=seed)
        L[i] = (c, capi.PDESampler(c, sp, projection="gather"), None)
    in_threads(make, lanes)
    out = {"solver": "hybridization" if is_hybrid(sp) else "saddle-point MINRES-BJ-GS",
           "workload": "EmbeddedPDESampler cube_tet_embed r=4, DoF " + str([lv.n_u + lv.n_s for lv in sp.levels[:3]]) +
                       f", original elements {[len(i) for i in sp.orig_index[:3]]}, per level nsamples x (Sample + Eval), {lanes} lanes",
           "levels": level_rates(L, (0, 1, 2), nrep=nrep, darcy=False)}
    for c, sm, _ in L:
        sm.close()
        c.close()
    if cpu and not is_hybrid(sp):
        try:
            out["cpu_baseline"] = cpu_level_rates(sp, None, seed, (0, 1, 2), (1, 2, 8))
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def config5(seed, probs, lanes=4, cpu=True, nrep=2):
    """BASELINE config 5 at full size on ONE GPU: SPE10-shaped box 1200 x 2200 x 170, 7 x 27 x 10 coarse cells refined 3 x
    (56 x 216 x 80 = 967 680 elements, 3.9 M Darcy DoF), L2ProjectionPDESampler on the box enlarged by one coarse cell per side,
    correlation length 100, 4 levels, k_ref == 1 (spe_perm.dat is not shipped).  probs: build_config5()."""
    from parelagmc_amd import capi, host_api
    sp, ops, dp = probs
    L = [None] * lanes

    def make(i):
        c = capi.Context(0, seed=seed)
        L[i] = (c, capi.PDESampler(c, sp, projection="l2", l2_ops=ops), capi.DarcySolver(c, dp))
    in_threads(make, lanes)
    out = {"workload": "SPE10-shaped box 56 x 216 x 80, L2ProjectionPDESampler (sampler DoF " + str([lv.n_u + lv.n_s for lv in sp.levels]) +
                       ") + DarcySolver (DoF " + str([lv.ndofs for lv in dp.levels]) + f"), 4 levels, algebraic Schur hierarchies "
                       f"on the stretched cells, per level nsamples x (Sample + Eval + SolveFwd), {lanes} lanes",
           "levels": level_rates(L, (0, 1, 2, 3), nrep=nrep, darcy=True)}
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


def timed_farm(farm, steps, first_step, warmup=2):
    """`warmup` untimed + `steps` timed steps of a SamplerFarm (device-synchronised on both sides);
    returns (seconds, stats of the timed steps)"""
    for i in range(warmup):
        farm.step(first_step + i)
    for lane in farm.lanes:
        lane[0].synchronize()
    t0 = time.perf_counter()
    st = []
    for i in range(steps):
        st += farm.step(first_step + warmup + i)
    for lane in farm.lanes:
        lane[0].synchronize()
    return time.perf_counter() - t0, st


def sampler_point(problem, dev, seed, nb, ns, steps, tag, what, cpu_per_core=None, opts=None, roofline=True,
                  cpu_problem=None, cpu_hybrid_problem=None):
    """One sampler-only figure on level 0 of `problem` (the harness shape of the headline: `ns` lanes x `nb` realizations per
    step): value, iterations, K5 roofline of the in-loop launches, and - cpu_per_core - the CPU column beside it."""
    farm = SamplerFarm(problem, dev, seed, nb, ns, opts=opts)
    nb = farm.nb
    dt, st = timed_farm(farm, steps, 0)
    check_stats(st, what)
    it = [t[0] for t in st]
    bi = float(sum(max(it[b:b + nb]) for b in range(0, len(it), nb)))
    L = problem.levels[0]
    zb = farm.lanes[0][1].z_bytes()
    out = {"workload": what, "value": steps * nb * ns / dt, "unit": "samples/s", "steps": steps, "batch": nb, "streams": ns,
           "mean_minres_iterations": float(np.mean(it)), "precond_storage": "fp32" if zb == 4 else "fp64"}
    if is_hybrid(problem):
        out["solver"] = "hybridization"
    if roofline and is_hybrid(problem):
        out["roofline"] = hybrid_roofline(farm, problem, nb, tag, (2 + steps + 1) * ns, bi, dt)
    elif roofline:
        out["roofline"] = operator_roofline(farm, problem, nb, tag, (2 + steps + 1) * ns,
                                            solver_bytes_per_iteration(problem, nb, zb), bi, dt)
    farm.close()
    if cpu_per_core:
        try:
            out["cpu_baseline"] = cpu_baseline(cpu_problem if cpu_problem is not None else problem, seed, cpu_per_core,
                                               hybrid_problem=cpu_hybrid_problem)
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def dropin_sampler(problem, dev, seed, n=16, use_graph=0):
    """The case the reference's UNCHANGED serial manager produces (/root/reference/src/MLMC_Manager.cpp:119-121): one
    Sample + one Eval per call, nbatch = 1, HOST pointers - exactly what INTEGRATION.md section 2 binds.  One handle."""
    from parelagmc_amd import capi
    c = capi.Context(dev, seed=seed)
    smp = capi.PDESampler(c, problem, capi.solver_opts(use_graph=use_graph, check_every=2))
    xi = smp.Sample(0, first_id=0, nbatch=1)
    for _ in range(2):
        smp.Eval(0, xi)
    t_s = t_e = 0.0
    its = []
    for i in range(n):
        t0 = time.perf_counter()
        xi = smp.Sample(0, first_id=100 + i, nbatch=1)
        t1 = time.perf_counter()
        _, st = smp.Eval(0, xi, return_stats=True)
        t2 = time.perf_counter()
        t_s += t1 - t0
        t_e += t2 - t1
        check_stats(st, "the one-realization-per-call path")
        its.append(st[0][0])
    smp.close()
    c.close()
    return {"samples_per_s": n / (t_s + t_e), "ms_per_Sample": 1e3 * t_s / n, "ms_per_Eval": 1e3 * t_e / n, "calls": n,
            "mean_minres_iterations": float(np.mean(its)), "use_graph": use_graph}


def darcy_hybrid_point(probs, dev, seed, launches=2):
    """The Darcy solve of config 3's finest level (cube_hex 64^3, 1 060 864 DoF) by both device solvers on the same lognormal
    draws: the default saddle-point MINRES with the block-diagonal preconditioner, and the hybridized solver
    (pmc_darcy_create_hybrid: the reference's "Hybridization" branch, /root/reference/src/DarcySolver.cpp:586,619).  One lane,
    `launches` full launches each after a warm-up launch; iterations and device-inclusive wall milliseconds per realization."""
    from parelagmc_amd import capi
    sp, dp = probs
    c = capi.Context(dev, seed=seed)
    smp = capi.PDESampler(c, sp)
    out = {"workload": f"Darcy SolveFwd, level 0 of config 3 ({dp.levels[0].n_u + dp.levels[0].n_p} DoF), one lane"}
    try:
        k = None
        for name, hyb in (("saddle_point", False), ("hybridized", True)):
            ds = capi.DarcySolver(c, dp, None, hybrid=hyb)
            try:
                w = ds.BatchWidth(0)
                if k is None:
                    k = smp.Eval(0, smp.Sample(0, first_id=11, nbatch=launches * w))
                ds.SolveFwd(0, k[:w])
                t0 = time.perf_counter()
                Q, _, st = ds.SolveFwd(0, k, return_stats=True)
                dt = time.perf_counter() - t0
                out[name] = {"ms_per_realization": dt * 1e3 / len(k), "iterations_mean": float(np.mean([t[0] for t in st])),
                             "all_converged": bool(all(t[1] == 1 for t in st)), "realizations": int(len(k)),
                             "per_launch": int(w), "Q0": float(Q[0])}
            finally:
                ds.close()
        out["Q_relative_difference"] = abs(out["hybridized"]["Q0"] - out["saddle_point"]["Q0"]) / abs(out["saddle_point"]["Q0"])
    finally:
        smp.close()
        c.close()
    return out


def dropin_mlmc(probs, dev, seed, ns=(2, 8, 32), use_graph=0):
    """Config 3 driven the way the reference's serial MLMC_Manager::InitRun drives its plugins
    (/root/reference/src/MLMC_Manager.cpp:113-173): per realization of a level pair Sample(l), Eval(l+1), SolveFwd(l+1),
    Eval(l, warm start), SolveFwd(l), one realization per call, host pointers."""
    from parelagmc_amd import capi
    sp, dp = probs
    c = capi.Context(dev, seed=seed)
    o = capi.solver_opts(use_graph=use_graph, check_every=2)
    smp, ds = capi.PDESampler(c, sp, o), capi.DarcySolver(c, dp, o)
    nl = 3

    def realization(l, rid):
        t = {}
        t0 = time.perf_counter()
        xi = smp.Sample(l, first_id=rid, nbatch=1)
        t["Sample"] = time.perf_counter() - t0
        qc = 0.0
        emb = None
        if l + 1 < nl:
            t0 = time.perf_counter()
            s, emb = smp.Eval(l + 1, xi, xi_level=l, want_embed=True)
            t["Eval_coarse"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            qc = ds.SolveFwd(l + 1, s)[0][0]
            t["SolveFwd_coarse"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        s = smp.Eval(l, xi, xi_level=l, init_s=emb, init_level=l + 1 if emb is not None else None, use_init=emb is not None)
        t["Eval"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        q = ds.SolveFwd(l, s)[0][0]
        t["SolveFwd"] = time.perf_counter() - t0
        return q - qc, t
    levels = []
    total_t = 0.0
    for l in (2, 1, 0):
        realization(l, 10 ** 6 + l)                      # warm-up: allocations
        acc = {}
        t0 = time.perf_counter()
        for i in range(ns[l]):
            _, t = realization(l, i)
            for k_, v in t.items():
                acc[k_] = acc.get(k_, 0.0) + v
        dt = time.perf_counter() - t0
        total_t += dt
        levels.append({"level": l, "realizations": ns[l], "realizations_per_s": ns[l] / dt,
                       "ms_per_call": {k_: 1e3 * v / ns[l] for k_, v in acc.items()}})
    ds.close()
    smp.close()
    c.close()
    cost = {e["level"]: 1.0 / e["realizations_per_s"] for e in levels}
    w = np.array([64.0, 256.0, 1024.0])
    return {"levels": levels, "use_graph": use_graph,
            "round_64_256_1024_realizations_per_s": float(w.sum() / sum(w[l] * cost[l] for l in range(3)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="realizations per plugin call and lane; 0 (default) = what the sampler prefers for the level "
                         "(pmc_sampler_batch_width: 64 up to 700 k rows, 32 up to ~5 M rows)")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent batches in flight per GPU (one context + host thread each): the launch-latency-"
                         "bound coarse-level kernels of one batch overlap the bandwidth-bound kernels of the other")
    ap.add_argument("--refine", type=int, default=5, help="uniform refinements of cube_tet (5 -> 595 968 DoF)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mlmc", action="store_true", help="skip the secondary config-3 (MLMC Darcy+SPDE) figure")
    ap.add_argument("--no-r6", action="store_true", help="skip the HBM-bound point (cube_tet r=6, 4.74 M DoF) under extra.r6")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline only: no extra.* at all (neither configs 3 / 4 / 5, r = 6, the hex point, the fp64 storage "
                         "run nor the drop-in path)")
    ap.add_argument("--all-configs", action="store_true",
                    help="configs 4 and 5 with their CPU columns as well (the default run measures their GPU legs only)")
    ap.add_argument("--only-config", type=int, choices=(4, 5), default=None,
                    help="of configs 4 and 5 run only this one")
    ap.add_argument("--inline-setup", action="store_true",
                    help="build the configurations' operators in this process instead of in background worker processes "
                         "(use it under rocprofv3: its preloaded library may have initialised the GPU, and a process that has "
                         "must not start another program)")
    ap.add_argument("--solver", choices=("hybrid", "saddle"), default="hybrid",
                    help="solver of the headline: hybrid = the reference's Hybridization option (element-local elimination to "
                         "the SPD multiplier system, MINRES + aggregation V-cycle; pmc_sampler_create_hybrid), saddle = its "
                         "default MINRES-BJ-GS on the saddle-point system (pmc_sampler_create).  The other one is measured "
                         "under extra.")
    ap.add_argument("--seed", type=int, default=20261003)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself and never touches the GPU
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a rank of a farm binds itself to the CPUs next to its GPU BEFORE torch / HIP start their threads
    affinity = pin_rank_to_gpu_numa(local_rank) if world > 1 else {"pinned": False, "why": "single rank"}

    # one-time finite-element setup of the configurations behind the headline: in worker processes, started before this
    # process initialises the GPU (they never touch it)
    # (--no-mlmc --no-r6 together: the headline-only command line of the older development scripts)
    extras = world == 1 and not args.no_extras and not (args.no_mlmc and args.no_r6)
    kinds = []
    if extras or (world > 1 and not args.no_mlmc and not args.no_extras):
        if not args.no_mlmc:
            kinds.append("c3")
    if extras:
        if not args.no_r6 and args.refine != 6:
            kinds.append("r6")
        for k in (4, 5):
            if args.only_config in (None, k):
                kinds.append(f"c{k}")
    pool = SetupPool([] if args.inline_setup else kinds)

    import torch
    import torch.distributed as dist

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

    problem = build_problem(args.refine)          # the saddle-point form: CPU baseline, extra.saddle_point_minres
    L = problem.levels[0]
    hybrid = args.solver == "hybrid"
    head = build_hybrid_problem(args.refine) if hybrid else problem
    dev = (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0
    red_dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    ns = max(1, args.streams)
    farm = SamplerFarm(head, dev, args.seed, args.batch, ns, world, rank)
    nb = farm.nb

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pool.wait()
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
            "id_stride": world, "cpu_affinity": affinity}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    out = None
    zb = farm.lanes[0][1].z_bytes()
    if rank == 0:
        next_batch = (args.warmup + args.steps + 1) * ns
        solver_text = (f"solver: hybridization (the reference's alternative to its default, src/PDESampler.cpp:291,307-311: "
                       f"element-local elimination to H lambda = G f on {head.levels[0].n_lambda} multipliers, nnz(H)="
                       f"{head.levels[0].H.nnz}; MINRES + one aggregation V-cycle; s = z f - G^T lambda) - the same field to "
                       f"the solver tolerance; the default MINRES-BJ-GS path is extra.saddle_point_minres"
                       if hybrid else "solver: MINRES-BJ-GS on the saddle-point system (the reference's default)")
        out = {
            "metric": "MC samples/sec (SPDE field + Darcy QoI) at stated DoF; SpMV HBM GB/s vs roofline",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"PDESampler cube_tet r={args.refine}, {L.n_u + L.n_s} DoF (n_s={L.n_s}, n_u={L.n_u}, "
                                   f"nnz(A)={L.nnz}), 1 MC level, corlen 0.1, SPDE field only (BASELINE config 2), "
                                   f"{ns} x {nb} realizations per step, all converged",
                       "workload_detail": "the Darcy QoI leg is measured on config 3 under extra.mlmc_config3; " + solver_text,
                       "tolerance": "MINRES 300 / rel 1e-6 / abs 1e-12",
                       "solver": "hybridization" if hybrid else "saddle-point MINRES-BJ-GS",
                       "mean_minres_iterations": acc[0] / max(acc[1], 1.0), "batch": nb, "streams": ns,
                       "parallelism": f"sample-farm x{world}",
                       # pmc_solver_opts.precond_storage of this run: operators, Krylov vectors, products, recurrences and
                       # ALL arithmetic are fp64 (`dtype`); fp32 is only how data INSIDE one application of the
                       # preconditioner is stored (extra.fp64_storage: the same run with everything stored fp64)
                       "precond_storage": "fp32" if zb == 4 else "fp64"},
            "devices": min(world, ndev), "ranks": per_rank,
            "roofline": (hybrid_roofline(farm, head, nb, f"r{args.refine}", next_batch, acc[2], dt) if hybrid else
                         operator_roofline(farm, problem, nb, f"r{args.refine}", next_batch,
                                           solver_bytes_per_iteration(problem, nb, zb), acc[2], dt)),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(problem, args.seed, 8, hybrid_problem=head if hybrid else None)
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
    farm.close()
    extra = {}

    def attempt(name, fn):
        """a secondary figure must never cost the headline line"""
        t_ = time.perf_counter()
        try:
            extra[name] = fn()
        except Exception as e:   # noqa: BLE001
            extra[name] = {"error": repr(e)}
        if isinstance(extra.get(name), dict):
            extra[name]["wall_s"] = time.perf_counter() - t_

    cpu = not args.no_cpu_baseline
    if extras:
        # everything fp64 (pmc_solver_opts.precond_storage = PMC_STORAGE_FP64): the same harness, a third of the steps
        if hybrid:
            # the reference's DEFAULT solver on the same workload: the headline of rounds 1-3, with its K5 roofline
            attempt("saddle_point_minres", lambda: sampler_point(
                problem, dev, args.seed, args.batch, ns, max(4, args.steps // 2), f"r{args.refine}",
                f"the headline workload with the reference's default solver: MINRES-BJ-GS on the saddle-point system "
                f"({L.n_u + L.n_s} DoF, nnz(A)={L.nnz}), {ns} lanes"))
        else:
            attempt("hybridization", lambda: sampler_point(
                build_hybrid_problem(args.refine), dev, args.seed, args.batch, ns, max(4, args.steps // 2), f"r{args.refine}",
                "the headline workload with the hybridized solver"))
        attempt("fp64_storage", lambda: sampler_point(
            head, dev, args.seed, args.batch, ns, max(4, args.steps // 3), f"r{args.refine}_fp64",
            f"the headline workload with precond_storage = fp64 (everything stored fp64, as the reference is: "
            f"/root/reference/src/PDESampler.cpp:279-333)", opts=capi.solver_opts(precond_storage=capi.PMC_STORAGE_FP64)))
        # the drop-in path of INTEGRATION.md section 2: one realization per call, host pointers
        attempt("dropin_nb1", lambda: {
            "what": "one Sample + one Eval (+ SolveFwd) per call with nbatch = 1 and HOST pointers, one handle: what the "
                    "reference's unchanged serial manager would drive (src/MLMC_Manager.cpp:113-173)",
            "config2": dropin_sampler(head, dev, args.seed, 16, 0),
            "config2_hipgraph": dropin_sampler(head, dev, args.seed, 16, 1),
            "config2_saddle_point_minres": dropin_sampler(problem, dev, args.seed, 16, 0),
            "config2_saddle_point_minres_hipgraph": dropin_sampler(problem, dev, args.seed, 16, 1)})
    if not args.no_mlmc and not args.no_extras:
        probs3 = pool.get("c3")
        if world == 1:
            def c3():
                m = mlmc_config3(args.seed, probs3, lanes=min(4, max(1, ns)), roofline=True)
                if cpu:
                    try:
                        m["cpu_baseline"] = darcy_cpu_baseline(probs3[0], probs3[1], args.seed)
                    except Exception as e:   # noqa: BLE001
                        m["cpu_baseline"] = {"error": repr(e)}
                return m
            attempt("mlmc_config3", c3)
            attempt("dropin_nb1_config3", lambda: dropin_mlmc(probs3, dev, args.seed))
            attempt("darcy_hybridization", lambda: darcy_hybrid_point(probs3, dev, args.seed))
            # the north star's "~1 M DoF 3D SPDE sampler": level 0 of config 3's hierarchy on its own, 1 060 864 DoF
            attempt("hex64", lambda: sampler_point(
                probs3[0], dev, args.seed, args.batch, ns, 6, "hex64",
                f"PDESampler cube_hex 64^3, {probs3[0].levels[0].n_u + probs3[0].levels[0].n_s} DoF, lognormal, level 0 of "
                f"config 3's hierarchy on its own: {ns} lanes", cpu_per_core=2 if cpu else None))
        else:
            # every rank takes part: sharded InitRun, accumulators summed by pmc_allreduce_sum_f64 (RCCL).  The
            # communicator is set up and tried FIRST, and the ranks agree (torch.distributed) on whether it works, so
            # that no rank enters the farm alone
            try:
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
                lanes3 = min(4, max(1, ns))
                if int(flag.item()) == 1:
                    m = mlmc_config3(args.seed, probs3, lanes=lanes3, farm=(world, rank, c0, dev))
                    m["collective"] = "rccl (pmc_allreduce_sum_f64 of the library's communicator)"
                    m["ranks_in_rccl_communicator"] = world
                elif dist.get_backend() == "gloo":
                    # rehearsal on a box with fewer GPUs than ranks: the same farm, the accumulators summed over gloo
                    def gloo_sum(buf):
                        tt = torch.from_numpy(buf)
                        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                    m = mlmc_config3(args.seed, probs3, lanes=lanes3, farm=(world, rank, c0, dev), reduce=gloo_sum)
                    m["collective"] = "gloo (rehearsal: the ranks share a GPU, RCCL refused: " + why + ")"
                else:
                    c0.close()
                    m = {"error": "RCCL communicator of the library not available on every rank: " + why}
                if "allreduce_ms" in m:
                    # every rank's time inside the round's one all-reduce (the slowest rank's wait is the others' cost)
                    ar = [None] * world
                    dist.all_gather_object(ar, m["allreduce_ms"])
                    m["allreduce_ms_per_rank"] = ar
                if rank == 0:
                    extra["mlmc_farm"] = m
            except Exception as e:   # noqa: BLE001 - the secondary figure must never cost the headline line
                if rank == 0:
                    extra["mlmc_farm"] = {"error": repr(e)}
    if extras:
        if args.only_config in (None, 4):
            def c4():
                sp4, hp4 = pool.get("c4")
                m = config4(args.seed, hp4 if hybrid else sp4, cpu=False, nrep=3)
                # the other solver on the same levels (tetrahedra with badly shaped cells: the block-diagonal preconditioner
                # of the saddle-point system needs 44-69 iterations there) and, with --all-configs, the CPU column
                m["other_solver"] = config4(args.seed, sp4 if hybrid else hp4, cpu=args.all_configs and cpu, nrep=3)
                return m
            attempt("c4", c4)
        if args.only_config in (None, 5):
            # hexahedra: the saddle-point form is the sparser one there (M couples 3 faces, H 11) and stays the faster solver
            attempt("c5", lambda: config5(args.seed, pool.get("c5"), cpu=args.all_configs and cpu, nrep=2))
        if not args.no_r6 and args.refine != 6:
            def r6():
                p6, h6 = pool.get("r6")
                what = ("PDESampler cube_tet r=6, 4743168 DoF: operator + vectors exceed the 256 MiB Infinity Cache, every "
                        "kernel is HBM-bound")
                m = sampler_point(h6 if hybrid else p6, dev, args.seed, args.batch, ns, 12, "r6", what,
                                  cpu_per_core=1 if cpu else None, cpu_problem=p6, cpu_hybrid_problem=h6 if hybrid else None)
                m["other_solver"] = sampler_point(p6 if hybrid else h6, dev, args.seed, args.batch, ns, 6, "r6", what)
                return m
            attempt("r6", r6)
        extra["setup_seconds_in_worker_processes"] = dict(pool.seconds)
    pool.close()
    if rank == 0 and extra:
        out["extra"] = extra
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["full_record"] = write_full_record(out)
        print(json.dumps(out), file=sys.stderr)          # the full record on stderr as well (a log keeps it)
        sys.stderr.flush()
        print(compact_line(out))                         # stdout: ONE line, < LINE_CAP bytes


if __name__ == "__main__":
    main()
