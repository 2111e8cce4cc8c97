#!/bin/bash
# development aid: batch width 16 against 32 on the large levels (config 2 headline, r = 6, config 3) -> gpurun_out/ab_width.txt
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/ab_width.txt
for b in 16 32; do
  for s in 4 2; do
    timeout -k 10 300 python $R/bench.py --batch $b --streams $s --steps $((640 / b / s)) --no-cpu-baseline --no-mlmc --no-r6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('c2 batch', $b, 'lanes', $s, 'value', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'frac', round(r['frac'], 3), 'solver', round(r['solver']['frac'], 3), flush=True)" >> $out || exit 1
  done
  timeout -k 10 300 python $R/bench.py --refine 6 --batch $b --streams $((64 / b)) --steps 4 --warmup 1 --no-cpu-baseline --no-mlmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('r6 batch', $b, 'lanes', $((64 / b)), 'value', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'frac', round(r['frac'], 3), flush=True)" >> $out || exit 1
done
for w in 300000 2000000; do
  echo "PMC_WIDE_ROWS=$w" >> $out
  PMC_WIDE_ROWS=$w timeout -k 10 200 python $R/scripts/c3_widths.py 4:256 2:256 >> $out 2>&1 || exit 1
done
cat $out
