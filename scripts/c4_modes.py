"""Coarsening mode comparison on the cube_tet_embed hierarchy (development aid)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), nref)
sp = build_sampler_problem(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=1)
ctx = capi.Context(0, seed=4)
for kw in (dict(mg_coarsening=0), dict(mg_coarsening=1)):
    smp = capi.PDESampler(ctx, sp, capi.solver_opts(**kw), projection="gather")
    xi = smp.Sample(0, first_id=0, nbatch=16)
    smp.Eval(0, xi)
    t0 = time.time()
    s, st = smp.Eval(0, xi, return_stats=True)
    dt = time.time() - t0
    print(kw, "iterations", sorted(set(t[0] for t in st)), f"{dt / 16 * 1e3:.2f} ms/realization", flush=True)
    smp.close()
