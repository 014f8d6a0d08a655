#!/bin/bash
# usage (here, after `gpurun -- 'bash tests/tools/round6_profiles.sh'`): tests/tools/sync_profiles6.sh <prefix, e.g. r6_20>
# copies the summaries of gpurun_out/r6p into profiles/<prefix>_* and rewrites profiles/traffic.json for the current kernel sources
set -e
cd "$(dirname "$0")/../.."
p=profiles/$1; r=gpurun_out/r6p
cp $r/bench_bcf.json ${p}_bench_bcf.json
cp $r/bench_bcf_lanes1.json ${p}_bench_bcf_lanes1.json
cp $r/bench_line_lanes1.json ${p}_bench_line_lanes1.json
cp $r/stats/p_kernel_stats.csv ${p}_kernel_stats_rocprofv3.csv
python3 tests/tools/short_stats.py ${p}_kernel_stats_rocprofv3.csv > ${p}_kernel_stats_short.txt
cp $r/stats_lanes1/p_kernel_stats.csv ${p}_kernel_stats_lanes1_rocprofv3.csv
python3 tests/tools/short_stats.py ${p}_kernel_stats_lanes1_rocprofv3.csv > ${p}_kernel_stats_lanes1_short.txt
cp $r/traffic/traffic_by_kernel.json ${p}_pmc_traffic_by_kernel.json
if [ -f $r/stats_bcf/p_kernel_stats.csv ]; then python3 tests/tools/short_stats.py $r/stats_bcf/p_kernel_stats.csv > ${p}_bcf_kernel_stats_lanes1_short.txt; fi
if [ -f $r/traffic_bcf/traffic_by_kernel.json ]; then cp $r/traffic_bcf/traffic_by_kernel.json ${p}_bcf_pmc_traffic_by_kernel.json; fi
if [ -f $r/stats_z/p_kernel_stats.csv ]; then python3 tests/tools/short_stats.py $r/stats_z/p_kernel_stats.csv > ${p}_stream_legs_kernel_stats_lanes1_short.txt; fi
[ -f $r/bgzf_ab.txt ] && cp $r/bgzf_ab.txt ${p}_bgzf_text_against_byte_level.txt
grep "^{" $r/c5_full.json > ${p}_c5_full_50000x100kb.json
( [ -f $r/gpu_tests.log ] && grep -n "passed\|failed" $r/gpu_tests.log; tail -1 $r/smoke.log ) > ${p}_gpu_tests_and_smoke.txt
[ -f $r/gpu_tests_size3_check.log ] && grep -n "passed\|failed" $r/gpu_tests_size3_check.log > ${p}_gpu_tests_size3_check.txt
python3 tests/tools/make_traffic_json.py $r/traffic/traffic_by_kernel.json 1000 1000000 46080 > /dev/null
python3 - "$r/bench_line.json" "${p}_bench_line.json" <<'PY'
import json, sys
t = json.load(open('profiles/traffic.json'))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if d['roofline'].get('traffic') is None:
    d['roofline']['traffic'] = t[d['roofline']['kernel']]['hbm_bytes_per_launch']   # (from the PMC passes of the same call; the bench run itself has none)
open(sys.argv[2], 'w').write(json.dumps(d) + "\n")
print({k: d[k] for k in ('value', 'ms_per_step', 'phase_ms')}, d['roofline'])
PY
