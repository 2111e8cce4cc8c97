#!/bin/bash
# V-cycle levels of the hybridized sampler: entries against SELL slots of S, S P and P^T (laboratory library, PMC_VERBOSE)
mkdir -p gpurun_out
HYB_LIB=libpmc_lab.so PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_prof.py ${R:-5} 2>&1 | grep "pmc\]" > gpurun_out/r4_hyb_levels.txt
cat gpurun_out/r4_hyb_levels.txt
