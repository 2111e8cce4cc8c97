#!/bin/bash
# single-lane kernel profile of the sampler solve on a small level (development aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profss -o ps --output-format csv -- python3 $R/scripts/sampler_prof.py ${1:-2} > $R/gpurun_out/profss.log 2>&1
rm -f $R/gpurun_out/profss/*kernel_trace.csv $R/gpurun_out/profss/*/*kernel_trace.csv
grep "^sampler\|dofs\|Error\|error" $R/gpurun_out/profss.log | head -5
