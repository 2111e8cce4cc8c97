#!/bin/bash
# round-4 GPU call 2: the new round-4 tests, the new default bench line (timed), a single-lane one-stream kernel profile with
# the laboratory library (every kernel alone on the chip: standalone rows of lincomb3 / wx / K5 / the V-cycle kernels)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "ratio_manager_cuts or bench_starts_its_own or abi_handshake or true_residual_of_the_sampler" > gpurun_out/r4_tests2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests2.log; tail -3 gpurun_out/r4_tests2.log
start=$(date +%s)
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err
echo "bench rc=$? wall $(( $(date +%s) - start )) s"
cp parelagmc_amd/lib/libpmc.so /tmp/libpmc_product.so
cp parelagmc_amd/lib/libpmc_lab.so parelagmc_amd/lib/libpmc.so
PMC_SPLIT_MIN=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof_s1 -o s1 -- python bench.py --streams 1 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r4_prof_s1.log 2>&1
echo "prof rc=$?"
cp /tmp/libpmc_product.so parelagmc_amd/lib/libpmc.so
find gpurun_out/r4_prof_s1 -name "*kernel_stats.csv" | head -3
