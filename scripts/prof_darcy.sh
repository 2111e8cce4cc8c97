#!/bin/bash
# single-stream kernel profile of the Darcy solve on hex 64^3 (development aid)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/profd -o pd --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/darcy_prof.py 4 > $GRAFT_REPO_ROOT/gpurun_out/profd.log 2>&1
