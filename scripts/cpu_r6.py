"""One-off CPU column of the HBM-bound point (cube_tet r = 6, 4.74 M DoF): the C restatement of the reference solver on
all host cores, two realizations per core (BASELINE.md, config 2')."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.cpu_baseline(bench.build_problem(6), 20261003, nsamples_per_core=2)), flush=True)
