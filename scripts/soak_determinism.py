"""Soak test (development aid): the same batch solved repeatedly on config 2 - two-stream schedule (late w / x update, split
Lanczos update, non-temporal streams) and one-stream schedule - must give bit-identical fields and iteration counts every
time, and the two schedules must agree with each other."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

nref = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
problem = bench.build_problem(nref)
n = problem.levels[0].n_s
ctx = capi.Context(0, seed=3)
xi = ctx.array(np.random.default_rng(1).standard_normal(16 * n))
ref = None
for mode in (1, 2):
    smp = capi.PDESampler(ctx, problem, capi.solver_opts(two_streams=mode))
    out = ctx.empty(16 * n)
    for r in range(reps):
        st = smp.Eval(0, xi, xi_level=0, s_out=out, return_stats=True)[-1]
        s = out.download()
        key = (s.tobytes(), tuple(t[0] for t in st))
        if ref is None:
            ref = key
        assert key == ref, f"mode {mode} repetition {r}: result differs"
        assert all(t[1] == 1 for t in st)
    print(f"two_streams={mode}: {reps} repetitions identical, iterations {sorted(set(ref[1]))}", flush=True)
    smp.close()
print("ok")
