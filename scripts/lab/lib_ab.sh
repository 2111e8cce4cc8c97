#!/bin/bash
# A/B of compile-time variants of libpmc.so: parelagmc_amd/lib/libpmc_<v>.so are swapped in on the GPU box's scratch copy
set -e
cd $GRAFT_REPO_ROOT
cp parelagmc_amd/lib/libpmc.so parelagmc_amd/lib/libpmc_base.so
for rep in 1 2; do
for v in base "$@"; do
  cp parelagmc_amd/lib/libpmc_$v.so parelagmc_amd/lib/libpmc.so
  for s in 1 4; do
  python bench.py --streams $s --steps 60 --warmup 5 --no-cpu-baseline --no-mlmc --no-r6 > gpurun_out/lib_ab.json 2> gpurun_out/lib_ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lib_ab.json").read().strip().splitlines()[-1])
print("$v streams=$s", round(d["value"],1), "K5 us", round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["roofline"]["frac"],3), "iso", round(d["roofline"]["isolated"]["frac"],3), "solver", round(d["roofline"]["solver"]["frac"],3))
PY
  done
  if [ $rep = 1 ] && [ -z "$NO_R6" ]; then
  python bench.py --refine 6 --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-mlmc --no-r6 > gpurun_out/lib_ab.json 2> gpurun_out/lib_ab.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lib_ab.json").read().strip().splitlines()[-1])
print("$v r6", round(d["value"],1), "K5 us", round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["roofline"]["frac"],3), "iso", round(d["roofline"]["isolated"]["frac"],3), "solver", round(d["roofline"]["solver"]["frac"],3))
PY
  fi
done
done
