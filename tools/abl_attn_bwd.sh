#!/bin/bash
# cost breakdown of the fused attention backward by timing ablations (MIVP_ATTN_BWD_ABL, see the kernel header)
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
for a in 0 1 2 3 4 5; do
  for sh in "" "shift"; do
    MIVP_ATTN_BWD_ABL=$a rocprofv3 --kernel-trace --stats --output-format csv -d $out/abl_${a}_$sh -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py enc0 20 $sh > $out/abl_${a}_$sh.log 2>&1
    f=$(ls $out/abl_${a}_$sh/*/*kernel_stats.csv | head -1)
    python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'bwd_fused' in r['Name']: print('ABL %s %-6s %-60s avg %8.1f us' % (sys.argv[2], sys.argv[3], r['Name'][:60], float(r['AverageNs'])/1e3))
" $f $a "$sh"
  done
done
