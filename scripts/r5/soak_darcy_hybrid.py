"""Soak (development aid): the hybridized Darcy solve (pmc_darcy_create_hybrid) on cube_hex 32^3, repeated alone and beside a
second lane that keeps the GPU busy with saddle-point solves: QoI bits and iteration counts must not change."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from parelagmc_amd import capi  # noqa: E402
from parelagmc_amd.fe import box_mesh, build_darcy_problem, build_hierarchy  # noqa: E402

h = build_hierarchy(box_mesh([4, 4, 4], [2, 2, 2], "hex"), 3)
dp = build_darcy_problem(h, [0, 1, 1, 1, 1, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1], n_mc_levels=1)
ctx = capi.Context(0, seed=3)
ds = capi.DarcySolver(ctx, dp, None, hybrid=True)
w = int(os.environ.get("SOAK_WIDTH", "0")) or ds.BatchWidth(0)
k = np.exp(np.random.default_rng(1).standard_normal((w, dp.levels[0].n_p)))
stop = False


def noise():
    c2 = capi.Context(0, seed=9)
    d2 = capi.DarcySolver(c2, dp)
    k2 = np.exp(np.random.default_rng(2).standard_normal((d2.BatchWidth(0), dp.levels[0].n_p)))
    while not stop:
        d2.SolveFwd(0, k2)
    d2.close()
    c2.close()


ref = None
for phase in ("alone", "beside another lane"):
    th = None
    if phase != "alone":
        th = threading.Thread(target=noise)
        th.start()
    for r in range(10):
        Q, _, st = ds.SolveFwd(0, k, return_stats=True)
        key = (Q.tobytes(), tuple(t[0] for t in st))
        if ref is None:
            ref = key
        assert key == ref, f"{phase}, repetition {r}: result differs"
        assert all(t[1] == 1 for t in st)
    print(f"{phase}: 10 repetitions of {w} realizations bit-identical, iterations {sorted(set(ref[1]))}", flush=True)
    if th:
        stop = True
        th.join()
ds.close()
ctx.close()
print("ok")
