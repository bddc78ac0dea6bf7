#!/bin/bash
# two-pass attention backward (head_dim 24 / 48: the dq + dkv pair) on the dec1 / dec0 block shapes: tools/ab_attn_2pass.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ab2_${tag}_$name -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py "$@" > $out/ab2_${tag}_$name.log 2>&1
  f=$(ls $out/ab2_${tag}_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print('%-60s calls %4s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
" $f
}
run d1 dec1 20
run d1s dec1 20 shift
run d0 dec0 20
run d0s dec0 20 shift
