#!/bin/bash
python3 scripts/sa_iso.py 32 0 3 || exit 1
python3 scripts/sa_iso.py 32 0 0 || exit 1
for p0 in 2 3; do for p1 in 3 4; do for w in 0 1; do
PMC_SA_ISO_PASSES0=$p0 PMC_SA_ISO_PASSES1=$p1 PMC_SA_ISO_WEAK=$w python3 scripts/sa_iso.py 32 1 0 || exit 1
done; done; done
