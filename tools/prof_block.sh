#!/bin/bash
# rocprofv3 kernel stats of one Swin block (tools/bench_block.py) per stage: tools/prof_block.sh <tag> stage...
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
for st in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pb_${tag}_$st -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py $st 10 $BLOCK_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pb_${tag}_$st.log 2>&1
  cp $GRAFT_REPO_ROOT/gpurun_out/pb_${tag}_$st/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/pb_${tag}_${st}.csv
done
