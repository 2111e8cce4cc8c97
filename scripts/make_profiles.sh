#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   0. gpurun_out/profile_stamp.json: sha256 of the libpmc.so and of the sources the passes below RUN (written here, on the
#      GPU box, at run time - scripts/collect_profiles.py copies it into profiles/pmc_traffic.json instead of hashing whatever
#      sits in the tree when it is called)
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the driver's)  -> gpurun_out/prof_bench/
#   1b. the headline alone (--no-extras: four lanes, hybridized), the kernel mix `value` is measured on -> gpurun_out/prof_l4/
#   2. the same for one lane per GPU (--streams 1): per-kernel time without overlap   -> gpurun_out/prof_s1/
#      (the headline's solver, hybridization) and gpurun_out/prof_s1_saddle/ (--solver saddle: MINRES-BJ-GS on the
#      saddle-point system, the K5 of extra.saddle_point_minres)
#   3. the same at the HBM-bound r = 6, one lane                                      -> gpurun_out/prof_s1_r6{,_saddle}/
#   4. the Darcy operator + M-block polynomial of config 3 in the MINRES loop, one lane (scripts/c3_darcy_op.py) -> gpurun_out/prof_c3/
#   5. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on short single-lane runs at r = 5 and r = 6 (both solvers) and on the
#      config-3 Darcy kernels                                   -> gpurun_out/pmc_{fetch,write}_{r5,r5s,r6,r6s,c3}/
#   6. (laboratory library present) one lane on ONE stream, every kernel alone on the chip: standalone rows of the flat
#      vector kernels (lincomb3, w / x update) against the streaming ceiling          -> gpurun_out/prof_s1_onestream/
# scripts/collect_profiles.py rNN then copies the summaries into profiles/ (tracked) and rebuilds profiles/pmc_traffic.json.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 - <<PY || exit 1
import json, sys
sys.path.insert(0, "$R")
import bench
json.dump({"libpmc_sha256": bench.lib_sha256(), "csrc_sha256": bench.csrc_sha256(), "where": "written on the GPU box by scripts/make_profiles.sh before its passes"},
          open("$R/gpurun_out/profile_stamp.json", "w"))
PY
run() {   # name, then the bench arguments
  local name=$1; shift
  rm -rf $R/gpurun_out/$name
  timeout -k 10 700 rocprofv3 "${PROF[@]}" -d $R/gpurun_out/$name -o p --output-format csv -- python3 $R/bench.py "$@" > $R/gpurun_out/$name.log 2>&1 || return 1
  python3 $R/scripts/by_grid.py $R/gpurun_out/$name $R/gpurun_out/${name}_by_grid.csv    # per (kernel, grid) rows, before the trace goes
  rm -f $R/gpurun_out/$name/*kernel_trace.csv $R/gpurun_out/$name/*/*kernel_trace.csv
  echo "$name done"
}
runpy() {   # name, script
  local name=$1; shift
  rm -rf $R/gpurun_out/$name
  timeout -k 10 500 rocprofv3 "${PROF[@]}" -d $R/gpurun_out/$name -o p --output-format csv -- python3 $R/$1 > $R/gpurun_out/$name.log 2>&1 || return 1
  rm -f $R/gpurun_out/$name/*kernel_trace.csv $R/gpurun_out/$name/*/*kernel_trace.csv
  echo "$name done"
}
# STAGE (a gpurun call lasts at most 20 minutes and every call starts from a fresh copy of the tree; gpurun_out/ is merged back
# after each): 1 = default bench + the headline alone; 2 = one lane at r = 5, both solvers, with their counter passes and the
# per-kernel roofline table; 3 = r = 6, config 3 and the one-stream laboratory pass; unset = everything in one go
STAGE=${STAGE:-all}
want() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }
pmc_pass() {   # suffix, bench arguments: the FETCH_SIZE and WRITE_SIZE passes of one configuration
  local suf=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    PROF=(--kernel-trace --pmc $c)
    n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    run pmc_${n}_$suf "$@" || return 1
  done
  PROF=(--kernel-trace --stats)
}
PROF=(--kernel-trace --stats)
if want 1; then
  if [ -z "$SKIP_BENCH_PROFILE" ]; then run prof_bench --steps 20 --warmup 5 --inline-setup || exit 1; fi
  run prof_l4 --steps 40 --warmup 5 --no-cpu-baseline --no-extras || exit 1     # the headline's own kernel mix: four lanes, hybridized
fi
if want 2; then
  run prof_s1 --streams 1 --steps 40 --no-cpu-baseline --no-extras || exit 1
  run prof_s1_saddle --solver saddle --streams 1 --steps 40 --no-cpu-baseline --no-extras || exit 1
  pmc_pass r5 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-extras || exit 1
  pmc_pass r5s --solver saddle --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-extras || exit 1
  # per-kernel roofline table of the one-lane hybridized profile (sizes from the library, durations and counters from the passes)
  python3 $R/scripts/roofline_table.py $R/gpurun_out/prof_s1_by_grid.csv $R/gpurun_out/pmc_fetch_r5_by_grid.csv $R/gpurun_out/pmc_write_r5_by_grid.csv > $R/gpurun_out/roofline_table.txt 2>&1 || exit 1
  echo "roofline table done"
fi
if want 3; then
  run prof_s1_r6 --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras || exit 1
  run prof_s1_r6_saddle --solver saddle --refine 6 --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras || exit 1
  pmc_pass r6 --refine 6 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-extras || exit 1
  pmc_pass r6s --solver saddle --refine 6 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-extras || exit 1
  runpy prof_c3 scripts/c3_darcy_op.py || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    PROF=(--kernel-trace --pmc $c)
    n=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    runpy pmc_${n}_c3 scripts/c3_darcy_op.py || exit 1
  done
  if [ -f $R/parelagmc_amd/lib/libpmc_lab.so ]; then
    PROF=(--kernel-trace --stats)
    # the laboratory library is SELECTED through the harness's loader override, never copied over the product's file
    PMC_LIB=$R/parelagmc_amd/lib/libpmc_lab.so PMC_SPLIT_MIN=0 run prof_s1_onestream --solver saddle --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras || exit 1
  fi
fi
