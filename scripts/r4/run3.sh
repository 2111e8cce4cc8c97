#!/bin/bash
# round-4 GPU call 3: (i) one lane on ONE stream with the laboratory library: standalone kernel rows; (ii) same-box A/B of
# the two-columns-in-flight K5 loop (libpmc_deep.so = -DPMC_K5_DEEP=1, 220 VGPRs, two waves per SIMD) against the product
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
L=parelagmc_amd/lib
cp $L/libpmc.so /tmp/libpmc_product.so
cp $L/libpmc_lab.so $L/libpmc.so
rm -rf gpurun_out/r4_prof_one
PMC_SPLIT_MIN=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof_one -o p --output-format csv -- python3 bench.py --streams 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r4_prof_one.log 2>&1
echo "prof rc=$?"
rm -f gpurun_out/r4_prof_one/*kernel_trace.csv gpurun_out/r4_prof_one/*/*kernel_trace.csv
out=gpurun_out/r4_deep_ab.txt
: > $out
for rep in 1 2; do
for v in product deep; do
  if [ $v = product ]; then cp /tmp/libpmc_product.so $L/libpmc.so; else cp $L/libpmc_$v.so $L/libpmc.so; fi
  for s in 4 1; do
    timeout -k 10 300 python bench.py --streams $s --steps $((s * 10)) --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$v lanes $s value', round(d['value'], 1), 'k5_us', round(r['avg_kernel_ms'] * 1e3, 2), 'frac', round(r['frac'], 3), 'iso', round(r['isolated']['frac'], 3), 'solver', round(r['solver']['frac'], 3), flush=True)" >> $out || exit 1
  done
done
done
cp /tmp/libpmc_product.so $L/libpmc.so
cat $out
