"""Darcy solve profile driver (development aid): hex 64^3, NB=16."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2 and sys.argv[2] == "tet":
    from parelagmc_amd.fe import kuhn_cube_tet
    h = build_hierarchy(kuhn_cube_tet(2.0), nref)
else:
    h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), nref)
if len(sys.argv) > 2 and sys.argv[2] == "tet":
    dp = build_darcy_problem(h, [0], [1], [1], n_mc_levels=1)      # one boundary attribute: pressure boundary all around
else:
    dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
print("dofs", dp.levels[0].n_u + dp.levels[0].n_p, flush=True)
ctx = capi.Context(0, seed=5)
ds = capi.DarcySolver(ctx, dp)
nb = int(__import__("os").environ.get("NB", "16"))
k = ctx.array(np.exp(np.random.default_rng(0).standard_normal(nb * dp.levels[0].n_p)))
ds.SolveFwd(0, k, nbatch=nb)
ctx.timer_start()
for _ in range(3):
    Q, C, st = ds.SolveFwd(0, k, nbatch=nb, return_stats=True)
ms = ctx.timer_stop() / 3
print(f"darcy {ms:.1f} ms/16, iters {np.mean([t[0] for t in st]):.1f}, {ms / np.mean([t[0] for t in st]) * 1e3:.0f} us/it")
