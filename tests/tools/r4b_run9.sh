cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
one() {
GDBAMD_LIB_PATH=$PWD/build/variants/$1/libgenomicsdb_amd.so python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 8 > gpurun_out/r4b/v2_$1_$2.json 2> gpurun_out/r4b/v2_$1_$2.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4b/v2_$1_$2.json').read().strip().splitlines()[-1])
    print('$1', round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms'].items()}, round(d['roofline']['avg_launch_ms'],2))
except Exception as e:
    print('$1 ERR', e); print(open('gpurun_out/r4b/v2_$1_$2.err').read()[-1500:])
PY
}
one base a; one cap192 a; one cap128 a; one cap192w4 a; one cap128w4 a; one base b; one cap192 b; one cap128 b; one cap128w4 b
for v in base cap128 cap128w4; do
GDBAMD_LIB_PATH=$PWD/build/variants/$v/libgenomicsdb_amd.so timeout 900 python -m pytest tests/test_many_input_alleles.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
done
