#!/bin/bash
# hybridized sampler: matching passes on the finest / lower levels (laboratory library), 4 lanes and 1 lane
mkdir -p gpurun_out
out=gpurun_out/r4_hybrid_passes.txt
: > $out
for pp in "3 3" "3 4" "3 2" "2 3" "4 4"; do
  set -- $pp
  echo "== passes0 $1 passes1 $2" >> $out
  HYB_LIB=libpmc_lab.so PMC_HYB_PASSES0=$1 PMC_HYB_PASSES1=$2 PMC_VERBOSE=1 timeout -k 10 300 python scripts/r4/hybrid_farm.py ${R:-5} hybrid 4,1 2>&1 | grep "hybrid sampler level\|lanes" | awk '!seen[$0]++' >> $out || exit 1
done
cat $out
