mkdir -p gpurun_out/r4b
for cfg in "0" "8" "16" "0" "8"; do
GDBAMD_SIZE3=$cfg python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 8 > gpurun_out/r4b/s3b_$cfg.json 2> gpurun_out/r4b/s3b_$cfg.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4b/s3b_$cfg.json').read().strip().splitlines()[-1])
    print('SIZE3=$cfg', round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms'].items()}, round(d['roofline']['avg_launch_ms'],2))
except Exception as e:
    print('SIZE3=$cfg ERR', e); print(open('gpurun_out/r4b/s3b_$cfg.err').read()[-1500:])
PY
done
GDBAMD_SIZE3_CHECK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_asm_paths.py tests/test_gpu_genome.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4b/gpu_tests_check2.txt
cat gpurun_out/r4b/gpu_tests_check2.txt
