set -x
mkdir -p gpurun_out/r4b
C3="python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 10000000 --window-bp 50000 --arena-mb 49152"
GDBAMD_BENCH_PIN=0 $C3 > gpurun_out/r4b/c3_pageable.json 2> gpurun_out/r4b/c3_pageable.err
GDBAMD_BENCH_PIN=1 $C3 > gpurun_out/r4b/c3_pinned.json 2> gpurun_out/r4b/c3_pinned.err
for r in 64 128 256; do
GDBAMD_RUN=$r python bench.py --no-stream --no-c3 --no-cpu-baseline --steps 6 > gpurun_out/r4b/head_run$r.json 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('ms_per_step'), d.get('phase_ms'), d.get('input_path'), d.get('positions_per_sec_device_only'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/r4b/*.err
