#!/bin/bash
# single-stream kernel profile of the bench workload (development aid): per-kernel time without stream overlap
cd /tmp && export TMPDIR=/tmp
for g in ${GRIDS:-512}; do
export PMC_DOT_GRID=$g
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1_$g -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --streams ${STREAMS:-1} --no-cpu-baseline --no-mlmc > $GRAFT_REPO_ROOT/gpurun_out/prof1_$g.log 2>&1 || exit 1
done
