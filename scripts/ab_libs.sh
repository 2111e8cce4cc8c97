#!/bin/bash
# A/B of prebuilt library variants on one GPU box (development aid): parelagmc_amd/lib/libpmc_<name>.so are copied over
# libpmc.so in turn (the box works on a scratch copy of the tree) and the default bench is run with each.
# usage: bash scripts/ab_libs.sh name [name ...]      -> gpurun_out/ab_libs.txt
R=${GRAFT_REPO_ROOT:-.}
L=$R/parelagmc_amd/lib
for v in "$@"; do
  cp $L/libpmc_$v.so $L/libpmc.so || exit 1
  timeout -k 10 300 python $R/bench.py --steps 40 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; m = d['extra']['mlmc_config3']; r6 = d['extra']['r6']
print('$v', 'c2', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), '| c3', round(m['realizations_per_s'], 1),
      'darcy_op_us', round(m['roofline']['avg_kernel_ms'] * 1e3, 2), 'darcy_mult_ms', [round(t['darcy_mult_ms'], 1) for t in m['phase_timers_ms']], '| r6', round(r6['value'], 1), 'k5_us',
      round(r6['roofline']['avg_kernel_ms'] * 1e3, 1), flush=True)
" >> $R/gpurun_out/ab_libs.txt || exit 1
done
cat $R/gpurun_out/ab_libs.txt
