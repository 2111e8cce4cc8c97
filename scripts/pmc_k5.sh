#!/bin/bash
# PMC diagnostics of the isolated K5 launches (development aid); ONE small counter group per pass, every pass under its own
# timeout (a group the hardware cannot collect aborts rocprofv3)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in ${GROUPS_OVERRIDE:-"GRBM_GUI_ACTIVE TA_BUSY_avr" "TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"}; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmck5_$i -o p --output-format csv -- python3 $R/scripts/spmv_probe.py 5 16 > $R/gpurun_out/pmck5_$i.log 2>&1 || { echo "group $i failed: $grp"; continue; }
  rm -f $R/gpurun_out/pmck5_$i/*kernel_trace.csv
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/pmck5_$i/p_counter_collection.csv")):
    if "0, false, 2>" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("$grp".split()[0] if False else k, "mean", sum(v) / len(v), "n", len(v))
PY
done
