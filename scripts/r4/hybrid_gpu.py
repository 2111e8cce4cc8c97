"""GPU check of the hybridized sampler (round 4): field against the saddle-point sampler on the same xi, iteration counts,
milliseconds per launch of 32 realizations.  Development aid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi  # noqa: E402
if os.environ.get("HYB_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "parelagmc_amd", "lib", os.environ["HYB_LIB"])
from parelagmc_amd.fe import (build_hierarchy, build_hybrid_sampler_problem, build_sampler_problem,  # noqa: E402
                              mesh_from_json)

mesh = sys.argv[2] if len(sys.argv) > 2 else "cube_tet"
for nref in [int(a) for a in sys.argv[1].split(",")]:
    h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", mesh + ".json")), nref)
    sp = build_sampler_problem(h, corlen=0.1, n_mc_levels=1)
    t0 = time.time()
    hp = build_hybrid_sampler_problem(h, corlen=0.1, n_mc_levels=1)
    t_h = time.time() - t0
    ctx = capi.Context(0, seed=7)
    nb = 32
    print(f"{mesh} r={nref}: n_u {sp.levels[0].n_u} n_s {sp.levels[0].n_s} nnz(A) {sp.levels[0].nnz} nnz(H) {hp.levels[0].H.nnz} "
          f"(hybrid setup {t_h:.1f} s)", flush=True)
    for tol in (1e-6, 1e-10):
        o = capi.solver_opts(rel_tol=tol)
        t0 = time.time()
        a = capi.PDESampler(ctx, sp, o)
        t_a = time.time() - t0
        t0 = time.time()
        b = capi.PDESampler(ctx, hp, o)
        t_b = time.time() - t0
        xi = a.Sample(0, 0, nb)
        sa, sta = a.Eval(0, xi, return_stats=True)
        sb, stb = b.Eval(0, xi, return_stats=True)
        err = np.linalg.norm(sa - sb, axis=1) / np.linalg.norm(sa, axis=1)
        # device-resident timing
        dx = ctx.array(xi)
        ds = ctx.empty(nb * a.SampleSize(0))
        res = {}
        for name, s in (("saddle", a), ("hybrid", b)):
            for _ in range(2):
                s.Eval(0, dx, xi_level=0, s_out=ds)
            ctx.synchronize()
            t0 = time.time()
            for _ in range(5):
                s.Eval(0, dx, xi_level=0, s_out=ds)
            ctx.synchronize()
            res[name] = (time.time() - t0) / 5 * 1e3
        print(f"  tol {tol:g}: iterations saddle {np.mean([t[0] for t in sta]):.1f} hybrid {np.mean([t[0] for t in stb]):.1f} "
              f"converged {all(t[1] for t in sta)} {all(t[1] for t in stb)} | field difference max {err.max():.2e} | "
              f"ms per 32: saddle {res['saddle']:.2f} hybrid {res['hybrid']:.2f} | create {t_a:.1f} / {t_b:.1f} s", flush=True)
        a.close()
        b.close()
    ctx.close()
