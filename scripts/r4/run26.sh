#!/bin/bash
# batched epilogue reads in the element-grouped Darcy kernels and the M-block polynomial: parity tests, then the default bench
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests2.txt 2>&1
rc=$?
tail -4 gpurun_out/r4_gpu_tests2.txt
[ $rc -eq 0 ] || exit $rc
bash scripts/r4/run15.sh
