#!/bin/bash
# launch-rate probe: config-3 MLMC rounds from one process vs two processes sharing the GPU (development aid)
python3 scripts/mlmc_lanes.py 4 > gpurun_out/tp_solo.log 2>&1 || exit 1
python3 scripts/mlmc_lanes.py 4 > gpurun_out/tp_a.log 2>&1 &
PA=$!
python3 scripts/mlmc_lanes.py 4 > gpurun_out/tp_b.log 2>&1 &
PB=$!
wait $PA || exit 1
wait $PB || exit 1
grep lanes gpurun_out/tp_solo.log gpurun_out/tp_a.log gpurun_out/tp_b.log
