#!/bin/bash
# same-box A/B (laboratory library): LDS tail of the hybridized V-cycle starting at the 4 964-row level (product: launches of
# more than 8 realizations) against one level further down for every launch width; one lane and four lanes, 32 and 64 wide
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for rep in 1 2; do
  for v in 8 256; do
    echo "== PMC_TAIL_LATER_NB=$v (rep $rep)"
    PMC_TAIL_LATER_NB=$v python scripts/r4/hybrid_farm.py 5 hybrid 1,4 32
  done
done
