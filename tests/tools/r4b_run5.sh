mkdir -p gpurun_out/r4b
for cfg in "0 0" "8 0" "8 1" "0 1" "8 1" "8 0"; do
set -- $cfg
GDBAMD_SIZE3=$1 GDBAMD_RES_LAYOUT=$2 python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 8 > gpurun_out/r4b/s3_$1_$2.json 2> gpurun_out/r4b/s3_$1_$2.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4b/s3_$1_$2.json').read().strip().splitlines()[-1])
    print('SIZE3=$1 LAYOUT=$2', round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms'].items()}, round(d['roofline']['avg_launch_ms'],2))
except Exception as e:
    print('SIZE3=$1 LAYOUT=$2 ERR', e); print(open('gpurun_out/r4b/s3_$1_$2.err').read()[-1500:])
PY
done
GDBAMD_SIZE3_CHECK=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4b/gpu_tests_check.txt
cat gpurun_out/r4b/gpu_tests_check.txt
