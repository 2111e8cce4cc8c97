#!/bin/bash
# bench on the final sources, then the profile passes
mkdir -p gpurun_out
bash scripts/r4/run15.sh || exit 1
bash scripts/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
rc=$?
tail -20 gpurun_out/make_profiles.log
exit $rc
