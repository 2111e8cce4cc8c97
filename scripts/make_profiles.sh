#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command               -> gpurun_out/prof_bench/
#   2. the same for one lane per GPU (--streams 1): per-kernel time without overlap -> gpurun_out/prof_s1/
#   3. the same at the HBM-bound r = 6, one lane                                    -> gpurun_out/prof_s1_r6/
#   4. the Darcy operator of config 3 in its MINRES loop, one lane (scripts/c3_darcy_op.py) -> gpurun_out/prof_c3/
#   5. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on short single-lane runs at r = 5, at r = 6 and on the config-3 Darcy
#      operator                                                         -> gpurun_out/pmc_{fetch,write}_{r5,r6,c3}/
# scripts/collect_profiles.py then copies the summaries into profiles/ (tracked) and rebuilds profiles/pmc_traffic.json.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {   # name, then the bench arguments
  local name=$1; shift
  timeout -k 10 500 rocprofv3 "${PROF[@]}" -d $R/gpurun_out/$name -o p --output-format csv -- python3 $R/bench.py "$@" > $R/gpurun_out/$name.log 2>&1 || return 1
  rm -f $R/gpurun_out/$name/*kernel_trace.csv $R/gpurun_out/$name/*/*kernel_trace.csv
  echo "$name done"
}
runpy() {   # name, script
  local name=$1; shift
  timeout -k 10 500 rocprofv3 "${PROF[@]}" -d $R/gpurun_out/$name -o p --output-format csv -- python3 $R/$1 > $R/gpurun_out/$name.log 2>&1 || return 1
  rm -f $R/gpurun_out/$name/*kernel_trace.csv $R/gpurun_out/$name/*/*kernel_trace.csv
  echo "$name done"
}
PROF=(--kernel-trace --stats)
run prof_bench || exit 1
run prof_s1 --streams 1 --steps 40 --no-cpu-baseline --no-mlmc --no-r6 || exit 1
run prof_s1_r6 --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-mlmc || exit 1
runpy prof_c3 scripts/c3_darcy_op.py || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  PROF=(--kernel-trace --pmc $c)
  n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  run pmc_${n}_r5 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-mlmc --no-r6 || exit 1
  run pmc_${n}_r6 --refine 6 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-mlmc || exit 1
  runpy pmc_${n}_c3 scripts/c3_darcy_op.py || exit 1
done
