#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command  -> gpurun_out/prof_bench/
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on a short single-stream bench -> gpurun_out/pmc_{fetch,write}/
# scripts/collect_profiles.py then copies the summaries into profiles/ (tracked).
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o b --output-format csv -- python3 $R/bench.py > $R/gpurun_out/prof_bench.log 2>&1 || exit 1
rm -f $R/gpurun_out/prof_bench/*kernel_trace.csv $R/gpurun_out/prof_bench/*/*kernel_trace.csv
echo "bench profile done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-mlmc > $R/gpurun_out/pmc_fetch.log 2>&1 || exit 1
rm -f $R/gpurun_out/pmc_fetch/*kernel_trace.csv $R/gpurun_out/pmc_fetch/*/*kernel_trace.csv
echo "fetch pass done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-mlmc > $R/gpurun_out/pmc_write.log 2>&1 || exit 1
rm -f $R/gpurun_out/pmc_write/*kernel_trace.csv $R/gpurun_out/pmc_write/*/*kernel_trace.csv
echo "write pass done"
