"""Soak (development aid): the hybridized sampler at config 2 - fused restriction, fp32 right-hand-side copy, 64 per launch, eight-
iteration w / x window - solved repeatedly, alone and with a second lane hammering the GPU beside it, must return bit-identical
fields and iteration counts every time; a realization's bits must not depend on what else runs."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

hp = bench.build_hybrid_problem(5)
n = hp.levels[0].n_s
ctx = capi.Context(0, seed=3)
smp = capi.PDESampler(ctx, hp)
w = int(os.environ.get("SOAK_WIDTH", "0")) or smp.BatchWidth(0)      # SOAK_WIDTH=1: the narrow-launch route (row-split levels, dense solve)
xi = ctx.array(np.random.default_rng(1).standard_normal(w * n))
out = ctx.empty(w * n)
stop = False


def noise():
    c2 = capi.Context(0, seed=9)
    s2 = capi.PDESampler(c2, hp)
    w2 = s2.BatchWidth(0)
    x2, o2 = c2.empty(w2 * n), c2.empty(w2 * n)
    i = 0
    while not stop:
        s2.Sample(0, first_id=i * w2, nbatch=w2, out=x2)
        s2.Eval(0, x2, xi_level=0, s_out=o2)
        i += 1
    s2.close()
    c2.close()


ref = None
for phase in ("alone", "beside another lane"):
    th = None
    if phase != "alone":
        th = threading.Thread(target=noise)
        th.start()
    for r in range(10):
        st = smp.Eval(0, xi, xi_level=0, s_out=out, return_stats=True)[-1]
        key = (out.download().tobytes(), tuple(t[0] for t in st))
        if ref is None:
            ref = key
        assert key == ref, f"{phase}, repetition {r}: result differs"
        assert all(t[1] == 1 for t in st)
    print(f"{phase}: 10 repetitions of {w} realizations bit-identical, iterations {sorted(set(ref[1]))}", flush=True)
    if th:
        stop = True
        th.join()
smp.close()
ctx.close()
print("ok")
