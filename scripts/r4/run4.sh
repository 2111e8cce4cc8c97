#!/bin/bash
# round-4 GPU call 4: the folded Lanczos term (operator product adds - beta v_old in its epilogue, update reads two vectors):
# parity tests with the product library, then a same-box A/B with the laboratory library (PMC_FOLD=0 switches it off)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "true_residual_of_the_sampler or sampler_matches_direct or two_stream_schedule or super_batches_match or warm_start or full_size_config2 or fp64_storage_reproduces or mc_manager or farm_of_two" > gpurun_out/r4_tests4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests4.log; tail -4 gpurun_out/r4_tests4.log
L=parelagmc_amd/lib
cp $L/libpmc.so /tmp/libpmc_product.so
cp $L/libpmc_lab.so $L/libpmc.so
out=gpurun_out/r4_fold_ab.txt
: > $out
for rep in 1 2 3; do
for f in 0 1; do
  for s in 4 1; do
    PMC_FOLD=$f timeout -k 10 300 python bench.py --streams $s --steps $((s * 10)) --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('fold $f lanes $s value', round(d['value'], 1), 'it', round(d['config']['mean_minres_iterations'], 2), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'solver', round(r['solver']['frac'], 3), flush=True)" >> $out || exit 1
  done
done
done
PMC_FOLD=1 timeout -k 10 300 python bench.py --refine 6 --streams 4 --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('r6 fold 1 value', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), flush=True)" >> $out
PMC_FOLD=0 timeout -k 10 300 python bench.py --refine 6 --streams 4 --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('r6 fold 0 value', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), flush=True)" >> $out
cp /tmp/libpmc_product.so $L/libpmc.so
cat $out
