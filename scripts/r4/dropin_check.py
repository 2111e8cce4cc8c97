"""one realization per call with host pointers (the drop-in path), both solvers, config 2 (round 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for name, p in (("hybrid", bench.build_hybrid_problem(5)), ("saddle", bench.build_problem(5))):
    r = bench.dropin_sampler(p, 0, 20261003, 16, 0)
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
