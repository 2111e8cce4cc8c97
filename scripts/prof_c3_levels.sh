#!/bin/bash
# per-level kernel profiles of the config-3 MLMC round, one lane (development aid): gpurun_out/profc3_L{0,1,2}_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ns in 64,0,0 0,256,0 0,0,1024; do
  C3_NS=$ns rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profc3_$i -o p --output-format csv -- python3 $R/scripts/c3_widths.py 1:256 > $R/gpurun_out/profc3_$i.log 2>&1
  f=$(ls $R/gpurun_out/profc3_$i/*kernel_stats.csv $R/gpurun_out/profc3_$i/*/*kernel_stats.csv 2>/dev/null | head -1)
  cp "$f" $R/gpurun_out/profc3_L${i}_stats.csv
  rm -rf $R/gpurun_out/profc3_$i
  i=$((i+1))
done
