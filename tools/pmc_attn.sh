#!/bin/bash
# PMC passes over one Swin block (tools/bench_block.py): each counter group in its own rocprofv3 run.
# usage: [BLOCK_ARGS="fwdonly shift"] tools/pmc_attn.sh <tag> [stage] [bench_block.py|bench_conv.py] ; results under gpurun_out/pmc_<tag>_<n>/
tag=$1; stage=${2:-dec2}; prog=${3:-bench_block.py}
cd /tmp; export TMPDIR=/tmp
n=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS"; do
  n=$((n+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n -- python $GRAFT_REPO_ROOT/tools/$prog $stage 2 $BLOCK_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n.log 2>&1 || echo "group $n failed"
done
