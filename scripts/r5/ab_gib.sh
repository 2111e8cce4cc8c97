#!/bin/bash
# laboratory library: both column groups of a slice in one workgroup for the post-smoothing kernel at 64 per launch (PMC_GIB, default 1)
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for rep in 1 2; do
  echo "== PMC_GIB=0 (rep $rep)"; PMC_GIB=0 python scripts/r4/hybrid_farm.py 5 hybrid 1,4 64
  echo "== PMC_GIB=1 (rep $rep)"; PMC_GIB=1 python scripts/r4/hybrid_farm.py 5 hybrid 1,4 64
done
