#!/bin/bash
# same-box A/B of compile-time variants (libraries built with make ... EXTRA=-DPMC_VC_MIN_WAVES=3 [...]): the fp32-gather (and
# fp64-gather) V-cycle kernels compiled for three wavefronts per SIMD instead of the 172-176 registers (two wavefronts) the
# compiler picks on its own.  cube_tet r = 5, hybridized, one and four lanes; then a kernel trace of six launches per library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for lib in libpmc_lab.so libpmc_w3.so libpmc_w3d.so; do
    echo "== $lib (rep $rep)"
    HYB_LIB=$lib python3 $R/scripts/r4/hybrid_farm.py 5 hybrid 1,4 32
  done
done
for lib in libpmc_lab.so libpmc_w3.so libpmc_w3d.so; do
  d=$R/gpurun_out/r5_prof_$lib
  rm -rf $d
  HYB_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace -d $d -o p --output-format csv -- python3 $R/scripts/r4/hybrid_prof.py 5 > $d.log 2>&1 || exit 1
  echo "== trace $lib"
  python3 $R/scripts/r4/trace_summary.py $(find $d -name '*kernel_trace.csv' | head -1) 70 14
  rm -rf $d
done
