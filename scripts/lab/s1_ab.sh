#!/bin/bash
# A/B of schedule switches (env var VAR, values 1 0 1 0) on the config-2 bench, one and four lanes
set -e
VAR=${1:-PMC_LATE_WX}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "two_stream or hipgraph or full_size_config2 or matches_direct" > gpurun_out/s1_test.log 2>&1 || { tail -30 gpurun_out/s1_test.log; exit 1; }
tail -2 gpurun_out/s1_test.log
for s in 1 4; do
for v in ${VALS:-1 0 1 0}; do
  env $VAR=$v python bench.py --streams $s --steps 60 --warmup 5 --no-cpu-baseline --no-mlmc --no-r6 > gpurun_out/s1_ab.json 2> gpurun_out/s1_ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/s1_ab.json").read().strip().splitlines()[-1])
print("streams=$s $VAR=$v", round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), round(d["roofline"]["solver"]["frac"],3))
PY
done
done
