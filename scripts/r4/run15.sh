#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_hyb.json 2> gpurun_out/r4_bench_hyb.err
rc=$?
echo "bench rc=$rc"; tail -5 gpurun_out/r4_bench_hyb.err
python scripts/r4/summarize_bench.py gpurun_out/r4_bench_hyb.json || true
exit $rc
