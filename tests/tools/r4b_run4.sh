mkdir -p gpurun_out/r4b
for v in 0 8 16 0 8; do
GDBAMD_SIZE3=$v python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 8 > gpurun_out/r4b/size3_$v.json 2> gpurun_out/r4b/size3_$v.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4b/size3_$v.json').read().strip().splitlines()[-1])
    print('SIZE3=$v', round(d['value']), round(d['ms_per_step'],2), d['phase_ms'])
except Exception as e:
    print('SIZE3=$v ERR', e); print(open('gpurun_out/r4b/size3_$v.err').read()[-1500:])
PY
done
GDBAMD_SIZE3_CHECK=1 timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4b/gpu_tests_check.txt
cat gpurun_out/r4b/gpu_tests_check.txt
