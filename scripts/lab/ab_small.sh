R=${GRAFT_REPO_ROOT:-.}
L=$R/parelagmc_amd/lib
out=$R/gpurun_out/ab_small.txt
for v in base lr base lr; do
  cp $L/libpmc_$v.so $L/libpmc.so || exit 1
  echo "== $v" >> $out
  C3_NS=0,0,2048 timeout -k 10 200 python $R/scripts/c3_widths.py 4:256 2>&1 | tail -1 >> $out
  C3_NS=0,512,0 timeout -k 10 200 python $R/scripts/c3_widths.py 4:256 2>&1 | tail -1 >> $out
  PMC_WIDE_ROWS=1000000 timeout -k 10 150 python $R/bench.py --batch 32 --streams 4 --steps 10 --no-cpu-baseline --no-mlmc --no-r6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c2 at 32 wide: value',round(d['value'],1),'k5_us',round(r['avg_kernel_ms']*1e3,1))" >> $out
done
cat $out
