#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-recipe --unique 2"
i=0
for c in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c -d gpurun_out/s12/p$i --output-format csv -- $B > gpurun_out/s12/p$i.log 2>&1 || echo "pass $i failed: $c"
done
echo done
