#!/bin/bash
# What bounds the gather kernels?  Hardware counters (separate rocprofv3 --pmc passes, a few counters each) over six
# hybridized launches of 32 realizations at cube_tet r = 5, one lane (scripts/r4/hybrid_prof.py); per (kernel, grid) means.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export HYB_LIB=${HYB_LIB:-libpmc_lab.so}
export PMC_TAIL_LATER_NB=${PMC_TAIL_LATER_NB:-8}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH TA_BUSY_avr TD_TD_BUSY_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
  i=$((i + 1))
  d=$R/gpurun_out/r5_pmc_probe_$i
  rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $d -o p --output-format csv -- python3 $R/scripts/r4/hybrid_prof.py 5 > $d.log 2>&1 || { echo "pass $i failed"; tail -5 $d.log; continue; }
  python3 - "$d" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
keys = sorted({c for v in acc.values() for c in v})
print("kernel,grid,launches," + ",".join(keys))
for k, v in sorted(acc.items(), key=lambda kv: -sum(len(x) for x in kv[1].values())):
    n = max(len(x) for x in v.values())
    if n < 20:
        continue
    print(f"\"{k[0]}\",{k[1]},{n}," + ",".join(f"{sum(v[c]) / max(len(v[c]), 1):.4g}" for c in keys))
PY
  rm -rf $d
done
