#!/bin/bash
# PMC diagnostics of the isolated K5 launches (development aid); one counter group per pass
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $GRAFT_REPO_ROOT/gpurun_out/pmck5_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/spmv_probe.py 5 16 > $GRAFT_REPO_ROOT/gpurun_out/pmck5_$i.log 2>&1 || exit 1
done
