#!/bin/bash
# K5 timing: warm/cold caches x with/without the fused dot x dot grid bound (development aid)
for f in ${FLUSH:-0 1024}; do for d in ${DOTS:-0 1}; do for g in ${GRIDS:-512}; do
echo "== flush ${f} MB, dot ${d}, grid ${g}"
PMC_PROBE_FLUSH_MB=$f PMC_PROBE_DOT=$d PMC_DOT_GRID=$g python3 scripts/spmv_probe.py 5 16 || exit 1
done; done; done
