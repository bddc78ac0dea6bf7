#!/bin/bash
# SQ counter passes over a few bench.py steps (every kernel of the step; tools/pmc_summary.py <tag> <kernel substring>).
# usage: tools/pmc_step.sh <tag> [workload]
tag=$1; wl=${2:-cfg1}
cd /tmp; export TMPDIR=/tmp
n=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"; do
  n=$((n+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --settle 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n.log 2>&1 || echo "group $n failed"
done
