#!/bin/bash
# usage (on the GPU box): tests/tools/prof_any.sh <outdir> <command...>  -> per-kernel stats of any command under gpurun_out/<outdir>
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o p -- "$@" </dev/null > /root/repo/$out/run.log 2>&1
f=$(find /root/repo/$out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then python3 /root/repo/tests/tools/short_stats.py "$f" | head -16; fi
tail -3 /root/repo/$out/run.log | cut -c1-300
