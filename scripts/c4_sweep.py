"""Solver-option grid on the cube_tet_embed hierarchy (BASELINE config 4's mesh; development aid)."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parelagmc_amd.fe import build_hierarchy, build_sampler_problem, mesh_from_json  # noqa: E402
from parelagmc_amd import capi  # noqa: E402
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = build_hierarchy(mesh_from_json(os.path.join(ROOT, "tests", "golden", "meshes", "cube_tet_embed.json")), nref)
sp = build_sampler_problem(h, corlen=0.1, embedded=True, lognormal=True, n_mc_levels=1)
ctx = capi.Context(0, seed=4)
n = sp.levels[0].n_s
nb = 16
rows = []
xi_d = None
for degM, sdeg, srat, gam in itertools.product((2, 3, 4), (2, 3), (8.0, 20.0), (0.7, 1.0, 1.4)):
    opts = capi.solver_opts(cheb_degree_M=degM, mg_smooth_degree=sdeg, mg_smooth_ratio=srat, schur_scale=gam)
    smp = capi.PDESampler(ctx, sp, opts, projection="gather")
    xi = smp.Sample(0, first_id=0, nbatch=nb)
    smp.Eval(0, xi)
    ctx.synchronize()
    ctx.timer_start()
    s, st = smp.Eval(0, xi, return_stats=True)
    ms = ctx.timer_stop()
    it = np.mean([t[0] for t in st])
    ok = all(t[1] == 1 for t in st)
    smp.close()
    rows.append((ms, degM, sdeg, srat, gam, it, ok))
    print(f"degM={degM} sdeg={sdeg} srat={srat} gamma={gam}: {ms:7.2f} ms/batch iters {it:5.1f} conv {ok}", flush=True)
rows.sort()
print("best:")
for r in rows[:8]:
    print(r)
