#!/bin/bash
# K5 isolated timing + FETCH_SIZE of the same launches (development aid)
R=$GRAFT_REPO_ROOT
python3 scripts/spmv_probe.py 5 16 1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/k5f -o p --output-format csv -- python3 $R/scripts/spmv_probe.py 5 16 > $R/gpurun_out/k5f.log 2>&1 || exit 1
rm -f $R/gpurun_out/k5f/*kernel_trace.csv
python3 - <<PY
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$R/gpurun_out/k5f/p_counter_collection.csv")) if "sell_spmm_kernel<16, 0, 0, false, 2" in r["Kernel_Name"]]
print("FETCH_SIZE raw KB mean", sum(v)/len(v), "n", len(v))
PY
