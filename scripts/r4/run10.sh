#!/bin/bash
# hybridized sampler: fp32 intermediates on aggregation levels; matching passes on the finest / lower levels
mkdir -p gpurun_out
out=gpurun_out/r4_hybrid2.txt
: > $out
for pp in "2 2" "3 2" "3 3" "2 3" "4 3"; do
  set -- $pp
  echo "== passes0 $1 passes1 $2" >> $out
  HYB_LIB=libpmc_lab.so PMC_HYB_PASSES0=$1 PMC_HYB_PASSES1=$2 PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_gpu.py 5 2>&1 | grep -v "aggregation level\|SA level\|sampler level 0: n_u" >> $out || exit 1
done
cat $out
