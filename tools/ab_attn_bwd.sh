#!/bin/bash
# A/B of the fused attention backward (env var toggles) on one box: tools/ab_attn_bwd.sh <tag> <ENVVAR>
tag=$1; var=$2
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {
  name=$1; envv=$2; shift 2
  if [ "$envv" != "-" ]; then export $envv; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/abb_${tag}_$name -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py "$@" > $out/abb_${tag}_$name.log 2>&1
  if [ "$envv" != "-" ]; then unset ${envv%%=*}; fi
  f=$(ls $out/abb_${tag}_$name/*/*kernel_stats.csv | head -1)
  echo "== $name"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']: print('%-70s calls %4s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
" $f
}
run new_p0 - enc0 20
run old_p0 $var=1 enc0 20
run new_p0s - enc0 20 shift
run old_p0s $var=1 enc0 20 shift
run new_n0 - enc0 20 noprompt
run old_n0 $var=1 enc0 20 noprompt
run new_n0s - enc0 20 noprompt shift
run old_n0s $var=1 enc0 20 noprompt shift
