#!/bin/bash
# A/B of the attention forward staging (LDS-DMA vs register path) on one box: tools/ab_attn_fwd.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {  # name, env assignment or "-", block args...
  name=$1; envv=$2; shift 2
  if [ "$envv" != "-" ]; then export $envv; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ab_${tag}_$name -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py "$@" > $out/ab_${tag}_$name.log 2>&1
  if [ "$envv" != "-" ]; then unset ${envv%%=*}; fi
  f=$(ls $out/ab_${tag}_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print('%-70s calls %4s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
" $f
}
run dma_f0 - enc0 10 fwdonly
run reg_f0 MIVP_ATTN_FWD_REG_STAGING=1 enc0 10 fwdonly
run dma_f0s - enc0 10 fwdonly shift
run reg_f0s MIVP_ATTN_FWD_REG_STAGING=1 enc0 10 fwdonly shift
run dma_p0 - enc0 10
run reg_p0 MIVP_ATTN_FWD_REG_STAGING=1 enc0 10
run dma_p0s - enc0 10 shift
run reg_p0s MIVP_ATTN_FWD_REG_STAGING=1 enc0 10 shift
