#!/bin/bash
mkdir -p gpurun_out
PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_prof.py 5 2>&1 | grep "pmc\]" > gpurun_out/r4_hyb_levels.txt
cat gpurun_out/r4_hyb_levels.txt
