cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for r in 64 128 256 64 128 256; do
GDBAMD_RUN=$r python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 8 > gpurun_out/r4b/run_$r.json 2> gpurun_out/r4b/run_$r.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4b/run_$r.json').read().strip().splitlines()[-1])
print('RUN=$r', round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms'].items()})
PY
done
