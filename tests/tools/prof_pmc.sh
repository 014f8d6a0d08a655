#!/bin/bash
# usage (GPU box): tests/tools/prof_pmc.sh <outdir> "<counters>" <bench args...>   (counters only: no trace domains besides kernel-trace)
out=gpurun_out/$1; shift
ctr="$1"; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /root/repo/$out -o p -- python /root/repo/bench.py "$@" --no-cpu-baseline </dev/null > /root/repo/$out/bench.log 2>&1
f=$(find /root/repo/$out -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then python3 /root/repo/tests/tools/pmc_sum.py "$f"; fi
tail -1 /root/repo/$out/bench.log | cut -c1-200
