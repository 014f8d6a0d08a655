#!/bin/bash
# page-assembly knobs at the c3 width (10 000 samples, 50 kb windows resident in HBM): ms per launch of k_assemble_write
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --samples 10000 --interval-bp 200000 --window-bp ${2:-50000} --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-c3 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']; print('step %.1f ms  write %.2f ms frac %.3f  phases %s' % (o['ms_per_step'], r['avg_launch_ms'], r['frac'], {k: round(v,1) for k,v in o['phase_ms'].items()}))"; }
run "X=0"
run "GDBAMD_RUN_W=16"
run "GDBAMD_RUN_W=64"
run "GDBAMD_ORDER_BLOCK_LOG2=10"
run "GDBAMD_ORDER_BLOCK_LOG2=14"
run "GDBAMD_ORDER_BLOCK_LOG2=16"
run "GDBAMD_WRITE_WAVES=4"
run "GDBAMD_XCD_AWARE=0"
run "X=1" 25000
run "X=2" 100000
