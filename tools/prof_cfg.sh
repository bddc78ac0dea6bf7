#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py for one workload: tools/prof_cfg.sh <workload> <tag> [extra bench args]
wl=$1; tag=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 3 --settle 0.3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-400
