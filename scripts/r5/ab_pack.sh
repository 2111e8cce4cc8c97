#!/bin/bash
# same-box A/B (laboratory library): multipliers renumbered so that the finest level's restriction is fused into its
# residual kernel (PMC_AGG_PACK=1, the product) against the separate product with P^T (0)
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for rep in 1 2; do
  for v in 0 1; do
    echo "== PMC_AGG_PACK=$v (rep $rep)"
    PMC_AGG_PACK=$v python scripts/r4/hybrid_farm.py 5 hybrid 1,4 32
  done
done
echo "== r = 6"
for v in 0 1; do
  echo "== PMC_AGG_PACK=$v"
  PMC_AGG_PACK=$v python scripts/r4/hybrid_farm.py 6 hybrid 4 32
done
