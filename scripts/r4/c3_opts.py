"""config 3 (MLMC Darcy + SPDE on hexes): V-cycle smoothing interval of sampler and Darcy hierarchies (round 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from parelagmc_amd import capi  # noqa: E402

probs = bench.build_config3()
for kw in (dict(), dict(mg_smooth_ratio=5.0), dict(mg_smooth_ratio=12.0), dict(mg_smooth_ratio=16.0), dict(mg_smooth_ratio=24.0)):
    m = bench.mlmc_config3(20261003, probs, lanes=4, opts=capi.solver_opts(**kw))
    ph = m["phase_timers_ms"]
    print(kw, f"{m['realizations_per_s']:.1f} realizations/s", [{k: round(v) for k, v in p.items()} for p in ph], flush=True)
