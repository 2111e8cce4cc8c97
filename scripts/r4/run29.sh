#!/bin/bash
# row-sorted S P / P^T on aggregation levels: hybrid tests, padding report, farm shape, one-lane profile
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -3 || exit 1
PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_prof.py 5 2>&1 | grep "V-cycle level" | cut -c1-200
timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4,1 > gpurun_out/r4_sorted_farm.txt 2>&1 || { tail -5 gpurun_out/r4_sorted_farm.txt; exit 1; }
cat gpurun_out/r4_sorted_farm.txt
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_hyb -o p -- python3 scripts/r4/hybrid_prof.py 5 > gpurun_out/r4_prof_hyb.log 2>&1
echo "prof rc=$?"
