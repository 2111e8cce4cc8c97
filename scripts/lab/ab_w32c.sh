# development aid: the default bench with the large sampler levels 16 against 32 realizations per launch, alternating on one box
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/ab_w32c.txt
for rep in 1 2 3; do
  for b in 16 32; do
    PMC_S_WIDE_ROWS=$([ $b = 32 ] && echo 5000000 || echo 300000) timeout -k 10 300 python $R/bench.py --batch $b --steps $((1920 / b)) --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; m = d['extra']['mlmc_config3']; r6 = d['extra']['r6']
print('width', $b, 'c2', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 1), 'frac', round(r['frac'], 3), 'solver', round(r['solver']['frac'], 3), '| c3', round(m['realizations_per_s'], 1), '| r6', round(r6['value'], 1), 'k5 frac', round(r6['roofline']['frac'], 3), flush=True)" >> $out
  done
done
cat $out
