#!/bin/bash
# bench throughput against the number of streams (development aid)
for s in ${STREAMS:-3 4 5 6 8}; do
python3 bench.py --no-cpu-baseline --no-mlmc --streams $s --steps 40 > gpurun_out/ss_$s.log 2>&1 || exit 1
tail -1 gpurun_out/ss_$s.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('streams', $s, 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))"
done
