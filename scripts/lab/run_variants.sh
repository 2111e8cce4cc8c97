#!/bin/bash
# runs the K5 laboratory binaries (baseline + tuning builds) on the exported config-2 operator
set -e
python - <<'PY'
import sys
sys.path.insert(0, 'scripts/lab')
import k5_lab
k5_lab.export(int(__import__('os').environ.get('LAB_REFINE', '5')), '/tmp/k5.bin')
PY
for v in "$@"; do
  echo "=== $v"
  LAB_CS=${LAB_CS:-none} timeout -k 10 300 parelagmc_amd/lib/$v /tmp/k5.bin ${LAB_NB:-16} 20
done
