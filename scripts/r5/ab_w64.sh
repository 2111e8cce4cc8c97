#!/bin/bash
# laboratory library: 64 realizations per launch on the 596 k-row level of config 2 (PMC_S_W64_ROWS=700000) against 32
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for rep in 1 2; do
  echo "== 32 per launch (rep $rep)"; python scripts/r4/hybrid_farm.py 5 hybrid 1,2,4 32
  echo "== 64 per launch (rep $rep)"; PMC_S_W64_ROWS=700000 python scripts/r4/hybrid_farm.py 5 hybrid 1,2,3,4 64
done
