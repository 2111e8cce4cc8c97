#!/bin/bash
# product library: lanes x realizations per launch of the hybridized sampler at config 2 (which split of the chip the headline should run)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  echo "== 64 per launch (rep $rep)"; python scripts/r4/hybrid_farm.py 5 hybrid 3,4,5,6 64
  echo "== 128 per launch (rep $rep)"; python scripts/r4/hybrid_farm.py 5 hybrid 2,3,4 128
done
