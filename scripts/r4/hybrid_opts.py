"""hybridized sampler, cube_tet r=5, 4 lanes x 32: V-cycle options (smoothing interval, coarse solve)  (round 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

hp = bench.build_hybrid_problem(5)
sets = {"ratio": (dict(mg_smooth_ratio=8.0), dict(mg_smooth_ratio=12.0), dict(mg_smooth_ratio=16.0), dict(mg_smooth_ratio=20.0),
                  dict(mg_smooth_ratio=30.0), dict(mg_smooth_ratio=50.0), dict(mg_smooth_ratio=100.0)),
        "first": (dict(), dict(mg_smooth_ratio=4.0), dict(mg_smooth_ratio=6.0), dict(mg_smooth_ratio=12.0), dict(mg_smooth_ratio=20.0),
                  dict(mg_coarse_degree=6), dict(mg_coarse_degree=20, mg_coarse_ratio=300.0))}
for kw in sets[sys.argv[1] if len(sys.argv) > 1 else "ratio"]:
    r = bench.sampler_point(hp, 0, 1234, 32, 4, 20, "x", "x", roofline=False, opts=capi.solver_opts(**kw))
    print(kw, f"{r['value']:.1f} samples/s, iterations {r['mean_minres_iterations']:.2f}", flush=True)
