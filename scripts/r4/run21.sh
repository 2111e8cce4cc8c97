#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu > gpurun_out/r4_hyb_tests2.txt 2>&1
rc=$?
tail -15 gpurun_out/r4_hyb_tests2.txt
[ $rc -eq 0 ] || exit $rc
bash scripts/r4/run15.sh
