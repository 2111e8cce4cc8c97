#!/bin/bash
# same-box A/B of prebuilt library variants on the hybridized config 2 (one and four lanes) + per-kernel trace of the top rows:
#   bash scripts/r5/ab_lib.sh libpmc_lab.so libpmc_tw.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    HYB_LIB=$lib python3 $R/scripts/r4/hybrid_farm.py 5 hybrid 1,4 32
  done
done
for lib in "$@"; do
  d=$R/gpurun_out/r5_prof_$lib
  rm -rf $d
  HYB_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace -d $d -o p --output-format csv -- python3 $R/scripts/r4/hybrid_prof.py 5 > $d.log 2>&1 || exit 1
  echo "== trace $lib"
  python3 $R/scripts/r4/trace_summary.py $(find $d -name '*kernel_trace.csv' | head -1) 70 16
  rm -rf $d
done
