#!/bin/bash
# usage (here, after `gpurun -- 'bash tests/tools/round4_profiles.sh'`): tests/tools/sync_profiles4.sh <prefix, e.g. r4_10>
# copies the summaries of gpurun_out/r4p into profiles/<prefix>_* and rewrites profiles/traffic.json for the current kernel sources
set -e
cd "$(dirname "$0")/../.."
p=profiles/$1; r=gpurun_out/r4p
cp $r/bench_bcf.json ${p}_bench_bcf.json
cp $r/stats/p_kernel_stats.csv ${p}_kernel_stats_rocprofv3.csv
python3 tests/tools/short_stats.py ${p}_kernel_stats_rocprofv3.csv > ${p}_kernel_stats_short.txt
cp $r/traffic/traffic_by_kernel.json ${p}_pmc_traffic_by_kernel.json
[ -f $r/bgzf_bench.txt ] && grep -v "amdgpu.ids" $r/bgzf_bench.txt > ${p}_bgzf_bench.txt
[ -f $r/type_stats.txt ] && cp $r/type_stats.txt ${p}_c3_width_window_type_stats.txt
( [ -f $r/gpu_tests.log ] && grep -n "passed\|failed" $r/gpu_tests.log; tail -1 $r/smoke.log ) > ${p}_gpu_tests_and_smoke.txt
[ -f $r/bgzf_sq_counters.txt ] && cp $r/bgzf_sq_counters.txt ${p}_bgzf_sq_counters.txt
[ -f $r/frag_bench.txt ] && cp $r/frag_bench.txt ${p}_fragment_load.txt
python3 tests/tools/make_traffic_json.py $r/traffic/traffic_by_kernel.json > /dev/null
python3 - "$r/bench_line.json" "${p}_bench_line.json" <<'PY'
import json, sys
t = json.load(open('profiles/traffic.json'))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d['roofline'].get('traffic') is None:
    d['roofline']['traffic'] = t[d['roofline']['kernel']]['hbm_bytes_per_launch']   # (from the PMC passes of the same call; the bench run itself has none)
open(sys.argv[2], 'w').write(json.dumps(d) + "\n")
print({k: d[k] for k in ('value', 'ms_per_step', 'phase_ms')}, d['roofline'])
PY
