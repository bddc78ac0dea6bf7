#!/bin/bash
# A/B of the shift-mask form in the attention forward (mask words vs byte classes): tools/ab_attn_mask.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {
  name=$1; envv=$2; shift 2
  if [ "$envv" != "-" ]; then export $envv; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/abm_${tag}_$name -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py "$@" > $out/abm_${tag}_$name.log 2>&1
  if [ "$envv" != "-" ]; then unset ${envv%%=*}; fi
  f=$(ls $out/abm_${tag}_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print('%-70s calls %4s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
" $f
}
run bits_f0s - enc0 20 fwdonly shift
run cls_f0s MIVP_ATTN_MASK_CLASSES=1 enc0 20 fwdonly shift
run bits_p0s - enc0 20 shift
run cls_p0s MIVP_ATTN_MASK_CLASSES=1 enc0 20 shift
run bits_d1s - dec1 20 shift
run cls_d1s MIVP_ATTN_MASK_CLASSES=1 dec1 20 shift
