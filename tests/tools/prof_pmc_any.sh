#!/bin/bash
# usage (GPU box): tests/tools/prof_pmc_any.sh <outdir> <kernel-regex> "<counters of pass 1>;<counters of pass 2>;..." <command...>
out=gpurun_out/$1; shift
re="$1"; shift
groups="$1"; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra G <<< "$groups"
for grp in "${G[@]}"; do
  d=/root/repo/$out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$re" --output-format csv -d $d -o p -- "$@" </dev/null > /root/repo/$out/run$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"(k_[a-z_0-9]+)",row["Kernel_Name"])
    if not m: continue
    k=m.group(1); agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
for k in sorted(agg):
    n=len(calls[k]); print("%-20s"%k,"disp",n,{c:"%.4g"%(v/n) for c,v in sorted(agg[k].items())})
PY
  rm -rf $d
  i=$((i+1))
done
