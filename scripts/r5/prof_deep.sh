#!/bin/bash
# per (kernel, grid) durations of six hybridized launches of 32 (cube_tet r = 5, one lane), laboratory library, for two
# settings of the deep gather loop / tail rule: where does a small V-cycle level spend its time?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export HYB_LIB=libpmc_lab.so
for cfg in "0 8" "2048 8" "2048 256"; do
  set -- $cfg
  d=$R/gpurun_out/r5_prof_deep_$1_$2
  rm -rf $d
  PMC_DEEP_WAVES=$1 PMC_TAIL_LATER_NB=$2 timeout -k 10 300 rocprofv3 --kernel-trace -d $d -o p --output-format csv -- python3 $R/scripts/r4/hybrid_prof.py 5 > $d.log 2>&1 || exit 1
  echo "== PMC_DEEP_WAVES=$1 PMC_TAIL_LATER_NB=$2"
  python3 $R/scripts/r4/trace_summary.py $(find $d -name '*kernel_trace.csv' | head -1) 70 40
  rm -rf $d
done
