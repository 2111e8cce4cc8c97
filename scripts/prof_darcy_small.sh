#!/bin/bash
# single-lane kernel profile of the Darcy solve on a small level (hex 16^3 by default; development aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profds -o pd --output-format csv -- python3 $R/scripts/darcy_prof.py ${1:-2} > $R/gpurun_out/profds.log 2>&1
rm -f $R/gpurun_out/profds/*kernel_trace.csv $R/gpurun_out/profds/*/*kernel_trace.csv
grep "^darcy\|dofs" $R/gpurun_out/profds.log
