#!/bin/bash
# usage (on the GPU box): tests/tools/prof_stats.sh <outdir> <bench args...>  -> per-kernel stats csv under gpurun_out/<outdir>
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o p -- python /root/repo/bench.py "$@" --no-cpu-baseline </dev/null > /root/repo/$out/bench.log 2>&1
f=$(find /root/repo/$out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then python3 /root/repo/tests/tools/short_stats.py "$f" | head -28; fi
tail -1 /root/repo/$out/bench.log | cut -c1-300
