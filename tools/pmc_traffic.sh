#!/bin/bash
# Fabric-side traffic of the two roofline kernels (bench.py `roofline.traffic` / `roofline_conv.traffic`): FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md: they do not fit one pass), over tools/bench_block.py enc0 fwdonly
# (stage-0 attention forward, un-shifted and shifted) and tools/bench_conv.py dec2.  usage: tools/pmc_traffic.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
pass() {  # name counter program args...
  name=$1; ctr=$2; shift 2
  timeout -k 5 120 rocprofv3 --pmc $ctr --output-format csv -d $out/traffic_${tag}_${name}_$ctr -- python3 "$@" > $out/traffic_${tag}_${name}_$ctr.log 2>&1 || echo "pass $name $ctr failed"
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  pass attn $ctr $GRAFT_REPO_ROOT/tools/bench_block.py enc0 4 fwdonly
  pass attns $ctr $GRAFT_REPO_ROOT/tools/bench_block.py enc0 4 fwdonly shift
  pass conv $ctr $GRAFT_REPO_ROOT/tools/bench_conv.py dec2 4
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic_summary.py $tag > $out/r03_traffic.json
cat $out/r03_traffic.json
