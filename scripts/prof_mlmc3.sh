#!/bin/bash
# kernel profile of the config-3 MLMC round (development aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BATCHES=32,32 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profm -o pm --output-format csv -- python3 $R/scripts/lab/mlmc3_ab.py $R > $R/gpurun_out/profm.log 2>&1
rm -f $R/gpurun_out/profm/*kernel_trace.csv $R/gpurun_out/profm/*/*kernel_trace.csv
