"""config-3 MLMC round throughput against the number of lanes (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import mlmc_config3
from parelagmc_amd import capi
opts = capi.solver_opts(use_graph=1) if os.environ.get('USE_GRAPH') else None
for lanes in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 12]:
    r = mlmc_config3(20261003, lanes=lanes, opts=opts)
    print(lanes, "lanes:", round(r["realizations_per_s"], 1), "realizations/s; s/sample/level", [round(x * 1e3, 3) for x in r["seconds_per_sample_per_level"]], flush=True)
