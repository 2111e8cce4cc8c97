cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C3_TWO_STREAMS=${C3_TWO_STREAMS:-0} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3 -o p --output-format csv -- python3 $R/scripts/c3_darcy_op.py > $R/gpurun_out/prof_c3.log 2>&1
f=$(ls $R/gpurun_out/prof_c3/*kernel_stats.csv $R/gpurun_out/prof_c3/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $R/gpurun_out/prof_c3_stats.csv
rm -rf $R/gpurun_out/prof_c3
