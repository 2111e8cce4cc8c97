#!/bin/bash
# final library: full GPU suite + smoke, default bench, profile passes
mkdir -p gpurun_out
bash scripts/r4/run19.sh || exit 1
bash scripts/r4/run15.sh || exit 1
SKIP_BENCH_PROFILE= bash scripts/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
rc=$?
tail -3 gpurun_out/make_profiles.log
exit $rc
