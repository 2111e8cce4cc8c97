#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the isolated K5 launches for every batch width (development aid)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/k5all_$c -o p --output-format csv -- python3 $R/scripts/spmv_probe.py 5 > $R/gpurun_out/k5all_$c.log 2>&1 || exit 1
rm -f $R/gpurun_out/k5all_$c/*kernel_trace.csv
done
python3 - <<PY
import csv, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("$R/gpurun_out/k5all_%s/p_counter_collection.csv" % c)):
        if "0, false, 2>" in r["Kernel_Name"]:
            nb = r["Kernel_Name"].split("<")[1].split(",")[0]
            acc[int(nb)].append(float(r["Counter_Value"]))
    for nb in sorted(acc):
        print(c, "nb", nb, "mean KB", sum(acc[nb]) / len(acc[nb]), "n", len(acc[nb]))
PY
