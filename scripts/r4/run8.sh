#!/bin/bash
# hybridized sampler on the GPU: first run
mkdir -p gpurun_out
timeout -k 10 600 python scripts/r4/hybrid_gpu.py 3,4,5 > gpurun_out/r4_hybrid.txt 2>&1
echo "rc $?" >> gpurun_out/r4_hybrid.txt
cat gpurun_out/r4_hybrid.txt
