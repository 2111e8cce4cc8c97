"""Config 3 probe: MLMC Darcy + SPDE sampler on cube_hex 64^3/32^3/16^3 (development aid)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd import capi, host_api  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.time()
h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), nref)
sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
print("fe setup", round(time.time() - t0, 1), "s; dofs", [L.ndofs for L in dp.levels], flush=True)
ctx = capi.Context(0, seed=5)
t0 = time.time()
smp = capi.PDESampler(ctx, sp)
ds = capi.DarcySolver(ctx, dp)
print("create", round(time.time() - t0, 1), "s", flush=True)
nb = 16
for lvl in range(3):
    n = sp.levels[lvl].n_s
    xi_d, s_d = ctx.empty(nb * n), ctx.empty(nb * n)
    smp.Sample(lvl, first_id=0, nbatch=nb, out=xi_d)
    smp.Eval(lvl, xi_d, xi_level=lvl, s_out=s_d)
    ctx.synchronize()
    ctx.timer_start()
    st = smp.Eval(lvl, xi_d, xi_level=lvl, s_out=s_d, return_stats=True)[1]
    ms_s = ctx.timer_stop()
    ds.SolveFwd(lvl, s_d, nbatch=nb)
    ctx.timer_start()
    Q, C, st2 = ds.SolveFwd(lvl, s_d, nbatch=nb, return_stats=True)
    ms_d = ctx.timer_stop()
    print(f"L{lvl}: sampler {ms_s:.1f} ms/16 ({np.mean([t[0] for t in st]):.0f} it)  darcy {ms_d:.1f} ms/16 "
          f"({np.mean([t[0] for t in st2]):.0f} it, conv {all(t[1] == 1 for t in st2)})  Q mean {Q.mean():.4f} std {Q.std():.3f}", flush=True)
    Q1, _ = ds.SolveFwd(lvl, np.ones((1, dp.levels[lvl].n_p)))
    print(f"     k=1: Q = {Q1[0]:.8f}", flush=True)
for nlanes in (1, 2, 4, 6):
    mgr = host_api.MLMCManager(3, sampler=smp, solver=ds, wall_time=True, batch=16)
    extra = []
    for _ in range(nlanes - 1):
        c2 = capi.Context(0, seed=5)
        extra.append((c2, capi.PDESampler(c2, sp), capi.DarcySolver(c2, dp)))
        mgr.add_lane(extra[-1][1], extra[-1][2])
    ns = [64, 256, 1024]
    mgr.InitRun([16 * nlanes] * 3)      # warm-up (allocations)
    mgr.Reset()
    t0 = time.time()
    r = mgr.InitRun(ns)
    dt = time.time() - t0
    print(f"lanes={nlanes}: InitRun {ns}: {dt:.2f} s -> {sum(ns) / dt:.1f} realizations/s; estimate {r['estimate']:.6f} "
          f"cost/sample {r['cost']} level_seconds {r['level_seconds']}", flush=True)
    mgr.close()
    for c2, s2, d2 in extra:
        d2.close(); s2.close(); c2.close()
