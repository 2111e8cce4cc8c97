#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the K5 laboratory kernels (development aid).  usage: pmc_lab.sh <refine> <name> [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ref=$1; name=$2; bin=${3:-k5_lab}
python3 - <<PY
import sys
sys.path.insert(0, '$R/scripts/lab'); sys.path.insert(0, '$R')
import k5_lab
k5_lab.export($ref, '/tmp/k5.bin')
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmclab_${name}_$c -o p --output-format csv -- $R/parelagmc_amd/lib/$bin /tmp/k5.bin 16 4 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmclab_${name}_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" $c $name <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "sell_spmm" in k or "lab_" in k:
        print(sys.argv[3], sys.argv[2], k[:80], len(v), round(sum(v) / len(v) / 1024.0, 1), "MB(raw)")
PY
  rm -rf $R/gpurun_out/pmclab_${name}_$c
done
