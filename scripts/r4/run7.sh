#!/bin/bash
# round-4 GPU call 7: fp64-gather kernels of a 32-wide batch as two column groups of 16 (libpmc_g16.so = -DPMC_LAB
# -DPMC_GROUP16=1; PMC_SPLIT16=1 switches the split on): same-box A/B + one-lane kernel rows + FETCH_SIZE of the two kernels
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
L=parelagmc_amd/lib
cp $L/libpmc.so /tmp/libpmc_product.so
cp $L/libpmc_g16.so $L/libpmc.so
python -m pytest tests -m gpu -x -q -k "sampler_matches_direct or true_residual_of_the_sampler" > gpurun_out/r4_tests7.log 2>&1; echo "pytest(split off) rc=$?"
PMC_SPLIT16=1 python -m pytest tests -m gpu -x -q -k "sampler_matches_direct or true_residual_of_the_sampler or full_size_config2" > gpurun_out/r4_tests7b.log 2>&1; echo "pytest(split on) rc=$?"; tail -3 gpurun_out/r4_tests7b.log
out=gpurun_out/r4_split16_ab.txt
: > $out
for rep in 1 2 3; do
for f in 0 1; do
  for s in 4 1; do
    PMC_SPLIT16=$f timeout -k 10 300 python bench.py --streams $s --steps $((s * 10)) --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('split16 $f lanes $s value', round(d['value'], 1), 'it', round(d['config']['mean_minres_iterations'], 2), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'solver', round(r['solver']['frac'], 3), flush=True)" >> $out || exit 1
  done
done
done
for f in 0 1; do
  rm -rf gpurun_out/r4_prof_g16_$f
  PMC_SPLIT16=$f PMC_SPLIT_MIN=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof_g16_$f -o p --output-format csv -- python3 bench.py --streams 1 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r4_prof_g16_$f.log 2>&1
  rm -f gpurun_out/r4_prof_g16_$f/*kernel_trace.csv gpurun_out/r4_prof_g16_$f/*/*kernel_trace.csv
  rm -rf gpurun_out/r4_pmc_g16_$f
  PMC_SPLIT16=$f PMC_SPLIT_MIN=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4_pmc_g16_$f -o p --output-format csv -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r4_pmc_g16_$f.log 2>&1
  rm -f gpurun_out/r4_pmc_g16_$f/*kernel_trace.csv gpurun_out/r4_pmc_g16_$f/*/*kernel_trace.csv
done
cp /tmp/libpmc_product.so $L/libpmc.so
cat $out
