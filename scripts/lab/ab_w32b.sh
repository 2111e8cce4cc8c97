R=${GRAFT_REPO_ROOT:-.}
cp $R/parelagmc_amd/lib/libpmc_lr.so $R/parelagmc_amd/lib/libpmc.so
out=$R/gpurun_out/ab_w32b.txt
for w in 300000 10000000 300000 10000000; do
  echo "PMC_WIDE_ROWS=$w" >> $out
  PMC_WIDE_ROWS=$w timeout -k 10 200 python $R/scripts/c3_widths.py 4:256 2>&1 | tail -1 >> $out
done
for w in 300000 10000000; do
  echo "config 4, PMC_WIDE_ROWS=$w" >> $out
  PMC_WIDE_ROWS=$w timeout -k 10 400 python $R/bench.py --all-configs --only-config 4 --steps 10 --no-r6 --no-mlmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(l['level'], l['realizations_per_launch'], round(l['realizations_per_s'],1)) for l in d['extra']['c4']['levels']])" >> $out
done
cat $out
