# development aid: 16 against 32 realizations per launch on the large levels over the lane count, library variant lr
R=${GRAFT_REPO_ROOT:-.}
cp $R/parelagmc_amd/lib/libpmc_lr.so $R/parelagmc_amd/lib/libpmc.so
out=$R/gpurun_out/ab_w32.txt
for cfg in "16 1" "32 1" "16 2" "32 2" "16 3" "32 3" "16 4" "32 4" "16 6" "32 6"; do
  set -- $cfg
  PMC_WIDE_ROWS=$([ $1 = 32 ] && echo 1000000 || echo 300000) timeout -k 10 150 python $R/bench.py --batch $1 --streams $2 --steps $((1280 / $1 / $2)) --no-cpu-baseline --no-mlmc --no-r6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5 width',$1,'lanes',$2,'value',round(d['value'],1))" >> $out
done
for cfg in "16 4" "32 2" "32 4" "32 1" "16 1"; do
  set -- $cfg
  PMC_WIDE_ROWS=$([ $1 = 32 ] && echo 10000000 || echo 300000) timeout -k 10 200 python $R/bench.py --refine 6 --batch $1 --streams $2 --steps $((128 / $1 / $2)) --warmup 1 --no-cpu-baseline --no-mlmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r6 width',$1,'lanes',$2,'value',round(d['value'],1))" >> $out
done
cat $out
