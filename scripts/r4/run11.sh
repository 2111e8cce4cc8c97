#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r4_hybrid_farm.txt
: > $out
HYB_LIB=libpmc_lab.so timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 saddle >> $out 2>&1 || exit 1
for pp in "3 3" "3 2" "2 2" "4 3"; do
  set -- $pp
  echo "== passes0 $1 passes1 $2" >> $out
  HYB_LIB=libpmc_lab.so PMC_HYB_PASSES0=$1 PMC_HYB_PASSES1=$2 timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid >> $out 2>&1 || exit 1
done
cat $out
