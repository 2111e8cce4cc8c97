"""BASELINE config 5 at full size on one GPU (development / robustness run, not part of the default bench):
SPE10-shaped box 1200 x 2200 x 170, 7 x 27 x 10 coarse cells refined 3 times (56 x 216 x 80 = 967 680 elements, 3.9 M DoF),
sampler on the box enlarged by one coarse cell per side (L2ProjectionPDESampler), correlation length 100, 4 levels,
stretched cells -> algebraic (smoothed-aggregation) Schur hierarchies.  Prints setup times, per-level MINRES iteration
counts and the MLMC round throughput."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd.fe import (box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem,  # noqa: E402
                              l2_projection_hierarchy)

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 3
setup_only = len(sys.argv) > 2 and sys.argv[2] == "setup"
t0 = time.time()
nx, ny, nz = 7, 27, 10
hx, hy, hz = 1200.0 / nx, 2200.0 / ny, 170.0 / nz
ho = build_hierarchy(box_mesh([nx, ny, nz], [1200.0, 2200.0, 170.0], "hex"), nref)
print(f"original hierarchy {[s.n_s for s in ho.spaces]} elements: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
he = build_hierarchy(box_mesh([nx + 2, ny + 2, nz + 2], [1200.0 + 2 * hx, 2200.0 + 2 * hy, 170.0 + 2 * hz], "hex",
                              origin=[-hx, -hy, -hz]), nref)
print(f"enlarged hierarchy {[s.n_s for s in he.spaces]} elements: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
sp = build_sampler_problem(he, corlen=100.0, lognormal=True)
ops = l2_projection_hierarchy(ho, he)
dp = build_darcy_problem(ho, [1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0])
print(f"operators: {time.time() - t0:.1f} s; sampler DoF {[L.n_u + L.n_s for L in sp.levels]}, "
      f"Darcy DoF {[L.ndofs for L in dp.levels]}", flush=True)
if setup_only:
    sys.exit(0)
from parelagmc_amd import capi, host_api  # noqa: E402
t0 = time.time()
ctx = capi.Context(0, seed=11)
smp = capi.PDESampler(ctx, sp, projection="l2", l2_ops=ops)
print(f"device sampler: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
ds = capi.DarcySolver(ctx, dp)
print(f"device Darcy solver: {time.time() - t0:.1f} s", flush=True)
nl = len(sp.levels)
for lvl in range(nl - 1, -1, -1):
    t0 = time.time()
    xi = smp.Sample(lvl, first_id=0, nbatch=4)
    s, st = smp.Eval(lvl, xi, return_stats=True)
    t1 = time.time()
    Q, C, st2 = ds.SolveFwd(lvl, s, return_stats=True)
    t2 = time.time()
    print(f"L{lvl}: sampler iters {[t[0] for t in st]} ({(t1 - t0) * 250:.1f} ms/sample)  darcy iters {[t[0] for t in st2]} "
          f"conv {[t[1] for t in st2]} ({(t2 - t1) * 250:.1f} ms/sample)  Q {np.round(Q, 3)}", flush=True)
mgr = host_api.MLMCManager(nl, sampler=smp, solver=ds, wall_time=True, batch=16)
ns = [16, 32, 64, 128][:nl]
t0 = time.time()
r = mgr.InitRun(ns)
dt = time.time() - t0
print(f"MLMC round {ns}: {dt:.2f} s, {sum(ns) / dt:.1f} realizations/s, estimate {r['estimate']:.4f}, "
      f"s/sample/level {np.round(r['level_seconds'] / np.array(ns), 4)}", flush=True)
