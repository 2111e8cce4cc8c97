#!/bin/bash
# kernel timeline of a short single-lane bench run (development aid): keeps the kernel trace, trimmed to the timed steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/trace -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --streams ${STREAMS:-1} --no-cpu-baseline --no-mlmc --no-r6 > $R/gpurun_out/trace.log 2>&1 || exit 1
f=$(find $R/gpurun_out/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" $R/gpurun_out/trace_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep a window of ~6000 kernels out of the middle of the run
mid = len(rows) // 4
sel = rows[mid:mid + 1500]
t0 = int(sel[0]['Start_Timestamp'])
with open(sys.argv[2], 'w') as o:
    o.write('start_us,dur_us,queue,stream,grid,wg,lds,vgpr,name\n')
    for r in sel:
        o.write('%.2f,%.2f,%s,%s,%s,%s,%s,%s,%s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                r.get('Queue_Id', ''), r.get('Stream_Id', ''), r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r['Kernel_Name'][:60].replace(',', ';')))
PY
rm -rf $R/gpurun_out/trace
