#!/bin/bash
# diagonal-last fused dot: parity tests, bench A/B (PMC_DIAG_LAST), FETCH_SIZE of the in-loop K5 at r5 / r6
set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu -k "block_operator or two_stream or full_size_config2 or matches_direct or hipgraph" > gpurun_out/dl_test.log 2>&1 || { tail -30 gpurun_out/dl_test.log; exit 1; }
tail -2 gpurun_out/dl_test.log
for s in 1 4; do
for v in 1 0 1 0; do
  PMC_DIAG_LAST=$v python bench.py --streams $s --steps 60 --warmup 5 --no-cpu-baseline --no-mlmc --no-r6 > gpurun_out/dl_ab.json 2> gpurun_out/dl_ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dl_ab.json").read().strip().splitlines()[-1])
print("streams=$s DIAG_LAST=$v", round(d["value"],1), "K5 us", round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["roofline"]["frac"],3), "solver", round(d["roofline"]["solver"]["frac"],3))
PY
done
done
for v in 1 0; do
  PMC_DIAG_LAST=$v python bench.py --refine 6 --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-mlmc --no-r6 > gpurun_out/dl_ab.json 2> gpurun_out/dl_ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dl_ab.json").read().strip().splitlines()[-1])
print("r6 DIAG_LAST=$v", round(d["value"],1), "K5 us", round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["roofline"]["frac"],3), "solver", round(d["roofline"]["solver"]["frac"],3))
PY
done
cd /tmp && export TMPDIR=/tmp
for r in 5 6; do
  rm -rf $R/gpurun_out/dl_pmc
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/dl_pmc -o p --output-format csv -- python3 $R/bench.py --refine $r --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-mlmc --no-r6 > $R/gpurun_out/dl_pmc.log 2>&1
  python3 - $R/gpurun_out/dl_pmc $r <<'PY'
import csv, os, sys, collections
acc = collections.defaultdict(list)
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("counter_collection.csv"):
            for row in csv.DictReader(open(os.path.join(root, f))):
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "sell_spmm_kernel<16, false, 0, true, 1" in k:
        print("r%s FETCH_SIZE x2 MB" % sys.argv[2], k[:70], len(v), round(2 * sum(v) / len(v) * 1024 / 1e6, 1))
PY
done
