"""config-3 MLMC round ([64, 256, 1024] realizations on 64^3 / 32^3 / 16^3) against lanes x realizations per plugin call
(development aid for the super-batch widths; width limits come from PMC_W64_ROWS / PMC_W128_ROWS / PMC_W256_ROWS).
usage: c3_widths.py lanes:batch [lanes:batch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi, host_api
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem

h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
ns = [int(x) for x in os.environ.get("C3_NS", "64,256,1024").split(",")]
okw = {}
if os.environ.get("C3_SMOOTH"):        # V-cycle smoothing degree (and optionally its interval ratio: "1" or "1:4")
    parts = os.environ["C3_SMOOTH"].split(":")
    okw["mg_smooth_degree"] = int(parts[0])
    if len(parts) > 1:
        okw["mg_smooth_ratio"] = float(parts[1])
opts = capi.solver_opts(**okw) if okw else None
for arg in sys.argv[1:] or ["4:32"]:
    lanes, batch = (int(x) for x in arg.split(":"))
    ctxs = [capi.Context(0, seed=20261003) for _ in range(lanes)]
    sm = [capi.PDESampler(c, sp, opts) for c in ctxs]
    dr = [capi.DarcySolver(c, dp, opts, hybrid=os.environ.get("C3_DARCY_HYBRID", "0") == "1") for c in ctxs]   # C3_DARCY_HYBRID=1: pmc_darcy_create_hybrid
    mgr = host_api.MLMCManager(3, sampler=sm[0], solver=dr[0], wall_time=True, batch=batch)
    for i in range(1, lanes):
        mgr.add_lane(sm[i], dr[i])
    mgr.InitRun(list(ns))      # warm-up: allocations at the widths of the timed round
    best = None
    for rep in range(2):
        mgr.Reset()
        t0 = time.perf_counter()
        r = mgr.InitRun(ns)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r)
    dt, r = best
    print(f"{lanes} lanes x {batch}: {sum(ns) / dt:8.1f} realizations/s  ({dt * 1e3:.0f} ms)  ms/sample/level",
          [round(float(x) * 1e3, 4) for x in r["cost"]], "estimate", round(float(r["estimate"]), 5), flush=True)
    mgr.close()
    for x in dr + sm:
        x.close()
    for c in ctxs:
        c.close()
