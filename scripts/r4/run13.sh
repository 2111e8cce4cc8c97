#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu -s > gpurun_out/r4_hyb_tests.txt 2>&1
rc=$?
tail -30 gpurun_out/r4_hyb_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4,6,8 > gpurun_out/r4_hyb_lanes.txt 2>&1 && cat gpurun_out/r4_hyb_lanes.txt
