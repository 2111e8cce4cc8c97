#!/bin/bash
# A/B of compile-time variants of libpmc.so on the Darcy fine-level solve and the config-3 MLMC round
cd $GRAFT_REPO_ROOT
cp parelagmc_amd/lib/libpmc.so parelagmc_amd/lib/libpmc_base.so
for rep in 1 2; do
for v in base "$@"; do
  cp parelagmc_amd/lib/libpmc_$v.so parelagmc_amd/lib/libpmc.so
  python scripts/darcy_prof.py 4 2>&1 | grep "^darcy" | sed "s/^/$v /"
  BATCHES=32,32 python scripts/lab/mlmc3_ab.py . 2>&1 | grep batch | sed "s/^/$v /" | head -2
done
done
