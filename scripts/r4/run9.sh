#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export HYB_LIB=libpmc_lab.so PMC_HYB_PASSES0=${P0:-3} PMC_HYB_PASSES1=${P1:-3} PMC_VERBOSE=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_hyb -o p -- python3 scripts/r4/hybrid_prof.py 5 > gpurun_out/r4_prof_hyb.log 2>&1
echo "prof rc=$?"
grep "hybrid sampler" gpurun_out/r4_prof_hyb.log | head -30
