#!/bin/bash
# A/B of compile-time variants of libpmc.so on the small-level Darcy solve (hex 16^3, 32 realizations), config 3 and config 2
cd $GRAFT_REPO_ROOT
cp parelagmc_amd/lib/libpmc.so parelagmc_amd/lib/libpmc_base.so
for rep in 1 2; do
for v in base "$@"; do
  cp parelagmc_amd/lib/libpmc_$v.so parelagmc_amd/lib/libpmc.so
  NB=32 python scripts/darcy_prof.py 2 2>&1 | grep "^darcy" | sed "s/^/$v /"
  BATCHES=32,32 python scripts/lab/mlmc3_ab.py . 2>&1 | grep batch | sed "s/^/$v /" | head -2
  python bench.py --streams 1 --steps 30 --no-cpu-baseline --no-mlmc --no-r6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v config2 one lane', round(d['value'],1))"
done
done
