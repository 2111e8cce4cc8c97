#!/bin/bash
mkdir -p gpurun_out
SKIP_BENCH_PROFILE= bash scripts/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
rc=$?
tail -20 gpurun_out/make_profiles.log
exit $rc
