#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python scripts/r4/hybrid_configs.py hex64,c4,c5 > gpurun_out/r4_hyb_configs.txt 2>&1
rc=$?
grep -v "^\[pmc\]" gpurun_out/r4_hyb_configs.txt | tail -30
exit $rc
