#!/bin/bash
# same-box A/B (laboratory library): deep gather loop on small V-cycle levels (PMC_DEEP_WAVES = launch size limit in
# wavefronts, 0 = off) x LDS tail one level later for every width (PMC_TAIL_LATER_NB=256) against the product rule (8)
cd "$(dirname "$0")/../.."
export HYB_LIB=libpmc_lab.so
for cfg in "0 8" "2048 8" "2048 256" "8192 8" "8192 256" "0 8"; do
  set -- $cfg
  echo "== PMC_DEEP_WAVES=$1 PMC_TAIL_LATER_NB=$2"
  PMC_DEEP_WAVES=$1 PMC_TAIL_LATER_NB=$2 python scripts/r4/hybrid_farm.py 5 hybrid 1,4 32
done
