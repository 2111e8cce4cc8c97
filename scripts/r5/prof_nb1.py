"""hybridized sampler, cube_tet r = 5, ONE realization per call (the drop-in path): the process rocprofv3 wraps (round 5)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

hp = bench.build_hybrid_problem(5)
ctx = capi.Context(0, seed=7)
b = capi.PDESampler(ctx, hp, capi.solver_opts(check_every=2))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
xi = b.Sample(0, 0, nb)
import time
for _ in range(3):
    b.Eval(0, xi)
t0 = time.perf_counter()
for _ in range(10):
    b.Eval(0, xi)
print("ms per Eval", 1e3 * (time.perf_counter() - t0) / 10)
