#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r4_hybrid_r6.txt
: > $out
for pp in "3 3" "2 3" "3 2" "2 2"; do
  set -- $pp
  echo "== passes0 $1 passes1 $2" >> $out
  HYB_LIB=libpmc_lab.so PMC_HYB_PASSES0=$1 PMC_HYB_PASSES1=$2 PMC_VERBOSE=1 timeout -k 10 400 python scripts/r4/hybrid_farm.py 6 hybrid 4 2>&1 | grep "hybrid sampler level\|lanes" | awk '!seen[$0]++' >> $out || exit 1
done
cat $out
