cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/parelagmc_amd/lib/libpmc_lr.so $R/parelagmc_amd/lib/libpmc.so
PMC_WIDE_ROWS=1000000 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof32 -o p --output-format csv -- python3 $R/bench.py --batch 32 --streams 1 --steps 20 --no-cpu-baseline --no-mlmc --no-r6 > $R/gpurun_out/prof32.log 2>&1
f=$(ls $R/gpurun_out/prof32/*kernel_stats.csv $R/gpurun_out/prof32/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $R/gpurun_out/prof32_stats.csv
rm -rf $R/gpurun_out/prof32
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof16 -o p --output-format csv -- python3 $R/bench.py --batch 16 --streams 1 --steps 40 --no-cpu-baseline --no-mlmc --no-r6 > $R/gpurun_out/prof16.log 2>&1
f=$(ls $R/gpurun_out/prof16/*kernel_stats.csv $R/gpurun_out/prof16/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $R/gpurun_out/prof16_stats.csv
rm -rf $R/gpurun_out/prof16
grep -o '"value": [0-9.]*' $R/gpurun_out/prof32.log | head -1; grep -o '"value": [0-9.]*' $R/gpurun_out/prof16.log | head -1
