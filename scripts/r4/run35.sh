#!/bin/bash
# hybridized sampler at r = 5: 64 realizations per launch (two column groups) against 32 (laboratory library: PMC_W64_ROWS)
mkdir -p gpurun_out
out=gpurun_out/r4_hyb_w64.txt
HYB_LIB=libpmc_lab.so timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4 32 > $out 2>&1 || exit 1
HYB_LIB=libpmc_lab.so PMC_W64_ROWS=700000 timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4,2,1 64 >> $out 2>&1 || exit 1
cat $out
