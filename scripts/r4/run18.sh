#!/bin/bash
mkdir -p gpurun_out
PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_prof.py 5 2>&1 | grep "V-cycle level" > gpurun_out/r4_hyb_levels2.txt
cat gpurun_out/r4_hyb_levels2.txt
timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4,1 > gpurun_out/r4_hyb_farm2.txt 2>&1 && cat gpurun_out/r4_hyb_farm2.txt
timeout -k 10 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -3
