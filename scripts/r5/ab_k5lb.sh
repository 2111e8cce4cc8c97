#!/bin/bash
# laboratory: K5 on H through the LDS-blocked product (PMC_K5_LB=1: every distinct gathered row loaded once into LDS per block
# of 256 rows) against the SELL gather kernel; field check, farm rates, kernel trace rows
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export HYB_LIB=libpmc_lab.so
PMC_K5_LB=1 python3 $R/scripts/r5/check_k5lb.py || exit 1
for rep in 1 2; do
  for v in 0 1; do
    echo "== PMC_K5_LB=$v"
    PMC_K5_LB=$v python3 $R/scripts/r4/hybrid_farm.py 5 hybrid 1,4 64 2>&1 | grep -v "^\[pmc\]"
  done
done
for v in 0 1; do
  d=$R/gpurun_out/r5_prof_k5lb_$v
  rm -rf $d
  PMC_K5_LB=$v timeout -k 10 300 rocprofv3 --kernel-trace -d $d -o p --output-format csv -- python3 $R/scripts/r4/hybrid_prof.py 5 > $d.log 2>&1 || exit 1
  echo "== trace PMC_K5_LB=$v"
  python3 $R/scripts/r4/trace_summary.py $(find $d -name '*kernel_trace.csv' | head -1) 70 10
  rm -rf $d
done
