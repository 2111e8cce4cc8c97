#!/bin/bash
# A/B of an environment switch on the Darcy fine-level solve (hex 64^3) and on the config-3 MLMC round
VAR=$1
for v in 1 0 1 0; do
  env $VAR=$v python scripts/darcy_prof.py 4 2>&1 | grep "^darcy" | sed "s/^/$VAR=$v /"
done
for v in 1 0; do
  env $VAR=$v BATCHES=32,32 python scripts/lab/mlmc3_ab.py . 2>&1 | grep batch | sed "s/^/$VAR=$v /" | head -2
done
