#!/bin/bash
# memory-path PMC passes over one Swin block (tools/bench_block.py); usage: tools/pmc_mem.sh <tag> [stage]
tag=$1; stage=${2:-dec2}
cd /tmp; export TMPDIR=/tmp
n=0
# at most two counters of one block per pass ("Request exceeds the capabilities of the hardware" aborts rocprofv3, which then
# hangs in its signal handler: every pass runs under its own short timeout)
for grp in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$((n+1))
  timeout -k 5 60 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n -- python $GRAFT_REPO_ROOT/tools/bench_block.py $stage 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n.log 2>&1 || echo "group $n failed"
done
