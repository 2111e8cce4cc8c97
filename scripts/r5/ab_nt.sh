#!/bin/bash
# laboratory library: non-temporal hints of the one-pass polynomial kernels (PMC_NT_POLY_MB, default 32) and of the in-loop
# operator (PMC_NT_MIN_MB, default 128) on the hybridized sampler at config 2, 64 per launch
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for rep in 1 2; do
  echo "== default (rep $rep)"; python scripts/r4/hybrid_farm.py 5 hybrid 1,4 64
  echo "== PMC_NT_POLY_MB=0 (rep $rep)"; PMC_NT_POLY_MB=0 python scripts/r4/hybrid_farm.py 5 hybrid 1,4 64
  echo "== PMC_NT_MIN_MB=100000 (rep $rep)"; PMC_NT_MIN_MB=100000 python scripts/r4/hybrid_farm.py 5 hybrid 1,4 64
done
