#!/bin/bash
# round-4 development aid: baseline of this round's box + env-only A/B of the sampler launch width (32 vs 64 columns per
# launch on config 2) and of the lane count -> gpurun_out/r4_ab1.txt
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/r4_ab1.txt
: > $out
run() {   # label, env assignments..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 300 python $R/bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$label', 'value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 2), 'it', round(d['config']['mean_minres_iterations'], 2), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'frac', round(r['frac'], 3), 'solver', round(r['solver']['frac'], 3), flush=True)" >> $out || exit 1
}
run "b32 s4" X=1 -- --batch 32 --streams 4 --steps 20
run "b32 s5" X=1 -- --batch 32 --streams 5 --steps 16
run "b32 s6" X=1 -- --batch 32 --streams 6 --steps 14
run "b32 s8" X=1 -- --batch 32 --streams 8 --steps 10
run "b64 s2" PMC_W64_ROWS=1000000 -- --batch 64 --streams 2 --steps 20
run "b64 s3" PMC_W64_ROWS=1000000 -- --batch 64 --streams 3 --steps 14
run "b64 s4" PMC_W64_ROWS=1000000 -- --batch 64 --streams 4 --steps 10
run "b32 s4 again" X=1 -- --batch 32 --streams 4 --steps 20
cat $out
