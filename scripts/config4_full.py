"""BASELINE config 4 at full size on one GPU (development / robustness run): EmbeddedPDESampler on cube_tet_embed.mesh,
4 uniform refinements (831 488 tets, 2.5 M DoF on the finest level), 3 Monte Carlo levels, log-normal, realizations of
the level pairs as MLMC_Manager drives them (coarse first, fine with the coarse field as initial guess)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.time()
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), nref)
sp = build_sampler_problem(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=3)
print(f"setup {time.time() - t0:.1f} s; elements {[L.n_s for L in sp.levels]}, DoF {[L.n_u + L.n_s for L in sp.levels]}, "
      f"original elements {[len(i) for i in sp.orig_index]}", flush=True)
from parelagmc_amd import capi  # noqa: E402
t0 = time.time()
ctx = capi.Context(0, seed=4)
smp = capi.PDESampler(ctx, sp, projection="gather")
print(f"device sampler: {time.time() - t0:.1f} s", flush=True)
nb = 16
for lvl in (2, 1, 0):
    xi = smp.Sample(lvl, first_id=0, nbatch=nb)
    smp.Eval(lvl, xi)                                   # warm-up (allocations)
    ctx.synchronize() if hasattr(ctx, "synchronize") else None
    t0 = time.time()
    if lvl < 2:
        sc, ec = smp.Eval(lvl + 1, xi, xi_level=lvl, want_embed=True)
        s, st = smp.Eval(lvl, xi, xi_level=lvl, init_s=ec, init_level=lvl + 1, use_init=True, return_stats=True)
    else:
        s, st = smp.Eval(lvl, xi, return_stats=True)
    dt = time.time() - t0
    print(f"L{lvl}: out size {s.shape[1]}, iterations {sorted(set(t[0] for t in st))}, converged {all(t[1] for t in st)}, "
          f"{dt / nb * 1e3:.2f} ms per realization (pair incl. host copies), mean log-field variance {np.log(s).var():.3f}", flush=True)
