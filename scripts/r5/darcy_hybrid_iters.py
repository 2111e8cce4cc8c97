"""Iterations and time of the Darcy solve on the levels of config 3 (cube_hex 64^3 / 32^3 / 16^3, lognormal coefficient from the
sampler): default (saddle-point, block-diagonal preconditioner) against the hybridized solver.  PMC_DARCY_HYB_SCALE (laboratory
library) sets the over-correction factor.  Development aid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy, build_sampler_problem  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get("MESH", "hex") == "tet":      # cube_tet with the boundary relabelled by position (inflow x = min, outflow x = max)
    from parelagmc_amd.fe import mesh_from_json
    m = mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet.json"))
    cen = m.verts[m.bdr].mean(axis=1)
    lo, hi = m.verts[:, 0].min(), m.verts[:, 0].max()
    m.bdr_attr = np.where(np.isclose(cen[:, 0], lo), 1, np.where(np.isclose(cen[:, 0], hi), 6, 2)).astype(m.bdr_attr.dtype)
    h = build_hierarchy(m, nref)
else:
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), nref)
sp = build_sampler_problem(h, corlen=0.1, lognormal=True, n_mc_levels=3)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=3)
print(f"{os.environ.get('MESH', 'hex')} refined {nref} x: Darcy DoF per level {[L.n_u + L.n_p for L in dp.levels[:3]]}", flush=True)
ctx = capi.Context(0, seed=20261003)
smp = capi.PDESampler(ctx, sp)
for hybrid in (False, True):
    t0 = time.perf_counter()
    ds = capi.DarcySolver(ctx, dp, None, hybrid=hybrid)
    setup = time.perf_counter() - t0
    for lvl in range(3):
        w = ds.BatchWidth(lvl) if hasattr(ds, "BatchWidth") else 16
        nb = 2 * w
        xi = smp.Sample(lvl, first_id=5, nbatch=nb)
        k = smp.Eval(lvl, xi)
        ds.SolveFwd(lvl, k[:w])
        t0 = time.perf_counter()
        Q, _, st = ds.SolveFwd(lvl, k, return_stats=True)
        dt = time.perf_counter() - t0
        its = [t[0] for t in st]
        print(f"{'hybridized' if hybrid else 'saddle-point'} level {lvl}: {nb} realizations ({w} per launch) {dt * 1e3 / nb:7.3f} ms each, "
              f"iterations {min(its)}-{max(its)} (mean {np.mean(its):.1f}), converged {all(t[1] == 1 for t in st)}, Q[0] {Q[0]:.8f}", flush=True)
    print(f"  (create: {setup:.1f} s)")
    ds.close()
smp.close()
ctx.close()
