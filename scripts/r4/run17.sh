#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python scripts/r4/hybrid_opts.py ratio > gpurun_out/r4_hyb_opts2.txt 2>&1 || { tail -5 gpurun_out/r4_hyb_opts2.txt; exit 1; }
cat gpurun_out/r4_hyb_opts2.txt
