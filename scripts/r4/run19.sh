#!/bin/bash
# full GPU suite + smoke + bench on the final sources
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests2.txt 2>&1
rc=$?
tail -5 gpurun_out/r4_gpu_tests2.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
