#!/bin/bash
# usage (here, after `gpurun -- 'bash tests/tools/round_profiles.sh'`): tests/tools/sync_profiles.sh <prefix, e.g. r2_20>
# copies the summaries of gpurun_out/r2p into profiles/<prefix>_* and rewrites profiles/traffic.json for the current kernel sources
set -e
cd "$(dirname "$0")/../.."
p=profiles/$1; r=gpurun_out/r2p
cp $r/bench_bcf.json ${p}_bench_bcf.json
cp $r/c5.txt ${p}_c5.txt
cp $r/stats/p_kernel_stats.csv ${p}_kernel_stats_rocprofv3.csv
python3 tests/tools/short_stats.py ${p}_kernel_stats_rocprofv3.csv > ${p}_kernel_stats_short.txt
cp $r/traffic/traffic_by_kernel.json ${p}_pmc_traffic_by_kernel.json
cp $r/store_bw.txt ${p}_store_bw.txt
if [ -f $r/bcfstats/p_kernel_stats.csv ]; then
  cp $r/bcfstats/p_kernel_stats.csv ${p}_bcf_kernel_stats_rocprofv3.csv
  python3 tests/tools/short_stats.py ${p}_bcf_kernel_stats_rocprofv3.csv > ${p}_bcf_kernel_stats_short.txt
fi
python3 tests/tools/make_traffic_json.py $r/traffic/traffic_by_kernel.json > /dev/null
python3 - "$r/bench_line.json" "${p}_bench_line.json" <<'PY'
import json, sys
t = json.load(open('profiles/traffic.json'))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d['roofline']['traffic'] = t[d['roofline']['kernel']]['hbm_bytes_per_launch']   # (the bench run itself has no PMC pass)
open(sys.argv[2], 'w').write(json.dumps(d) + "\n")
print({k: d[k] for k in ('value', 'ms_per_step', 'phase_ms')}, d['roofline'])
PY
