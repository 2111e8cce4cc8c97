"""one launch of 16 hybridized Darcy solves on cube_hex 64^3 (kernel trace target; development aid)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy  # noqa: E402

h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 4)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
ctx = capi.Context(0, seed=20261003)
ds = capi.DarcySolver(ctx, dp, None, hybrid=os.environ.get("SADDLE", "0") != "1")
k = np.exp(np.random.default_rng(3).standard_normal((int(os.environ.get("NBATCH", "16")), dp.levels[0].n_p)))
ds.SolveFwd(0, k)
import time
t0 = time.perf_counter()
Q, _, st = ds.SolveFwd(0, k, return_stats=True)
print("iterations", [t[0] for t in st], "ms per call", (time.perf_counter() - t0) * 1e3)
ds.close()
ctx.close()
