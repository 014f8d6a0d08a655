#!/bin/bash
# usage (on the GPU box): tests/tools/prof_sq.sh <outdir> <kernel-regex> <bench args...>
# SQ instruction-mix / occupancy counters of the kernels matching the regex, one rocprofv3 --pmc pass per group of counters
# (separate passes, --kernel-trace only), summed per kernel and counter (pmc_sum.py), divided by the number of dispatches
out=gpurun_out/$1; shift
re="$1"; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA"; do
  d=/root/repo/$out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$re" --output-format csv -d $d -o p -- python /root/repo/bench.py "$@" --no-cpu-baseline --no-stream </dev/null > /root/repo/$out/run$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 /root/repo/tests/tools/pmc_sum.py "$f" 999 per-dispatch
  i=$((i+1))
done
