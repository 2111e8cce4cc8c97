#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python scripts/r4/hybrid_farm.py 5 hybrid 4,5,6,8 > gpurun_out/r4_hyb_lanes2.txt 2>&1 && cat gpurun_out/r4_hyb_lanes2.txt
timeout -k 10 900 python scripts/r4/hybrid_configs.py hex64,c5 > gpurun_out/r4_hyb_configs2.txt 2>&1
rc=$?
grep -v "^\[pmc\]" gpurun_out/r4_hyb_configs2.txt | tail -12
exit $rc
