#!/bin/bash
# all workloads of DESIGN.md section 6 on one box
run() { timeout -k 10 200 python bench.py --steps 30 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %8.3f ms %8.1f vol/s roofline %.3f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], (d.get('roofline_conv') or {}).get('frac') or 0))" "$@"; }
run --workload cfg1
run --workload cfg2
run --workload cfg3
run --workload cfg1 --window 8,8,4
run --workload cfg2 --window 8,8,4
run --workload cfg3 --window 8,8,4
run --workload sup_all
run --workload cfg1 --dropout 0.1
run --workload cfg2 --dropout 0.1
run --workload cfg0
run --workload cfg4
run --workload cfg4 --fp8-attn
run --workload cfg1
