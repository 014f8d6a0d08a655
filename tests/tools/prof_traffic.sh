#!/bin/bash
# usage (GPU box): tests/tools/prof_traffic.sh <outdir> <bench args...>
# Two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, then per-kernel sums.
out=gpurun_out/$1; shift
mkdir -p $out/fetch $out/write
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /root/repo/$out/fetch -o p -- python /root/repo/bench.py "$@" --no-cpu-baseline </dev/null > /root/repo/$out/fetch/bench.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /root/repo/$out/write -o p -- python /root/repo/bench.py "$@" --no-cpu-baseline </dev/null > /root/repo/$out/write/bench.log 2>&1
python3 /root/repo/tests/tools/traffic_sum.py /root/repo/$out
