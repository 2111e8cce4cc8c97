#!/bin/bash
# round-4 GPU call 5: the full GPU suite, smoke, the default bench line on the final sources
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests5.log; tail -3 gpurun_out/r4_tests5.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke5.log 2>&1; echo "smoke rc=$?"
start=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench5.json 2> gpurun_out/r4_bench5.err
echo "bench rc=$? wall $(( $(date +%s) - start )) s"
