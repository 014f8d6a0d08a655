mkdir -p gpurun_out/r4b
C3="python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 10000000 --window-bp 50000 --arena-mb 49152"
GDBAMD_STAGE_PRIORITY=0 GDBAMD_STAGE_SUB_MB=256 $C3 > gpurun_out/r4b/c3_p0_s256.json 2> gpurun_out/r4b/c3_p0_s256.err
GDBAMD_STAGE_PRIORITY=1 GDBAMD_STAGE_SUB_MB=256 $C3 > gpurun_out/r4b/c3_p1_s256.json 2> gpurun_out/r4b/c3_p1_s256.err
GDBAMD_STAGE_PRIORITY=0 GDBAMD_STAGE_SUB_MB=1024 $C3 > gpurun_out/r4b/c3_p0_s1024.json 2> gpurun_out/r4b/c3_p0_s1024.err
GDBAMD_STAGE_PRIORITY=1 GDBAMD_STAGE_SUB_MB=1024 $C3 > gpurun_out/r4b/c3_p1_s1024.json 2> gpurun_out/r4b/c3_p1_s1024.err
GDBAMD_STAGE_PRIORITY=0 GDBAMD_STAGE_SUB_MB=256 GDBAMD_STAGE_BUDGET_MB=4096 $C3 > gpurun_out/r4b/c3_p0_s256_b4096.json 2> gpurun_out/r4b/c3_p0_s256_b4096.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/c3_p*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ip=d['input_path']
        print(f, round(d['value']), 'wall', round(ip['wall_s'],2), 'stage', round(ip['t_stage_s'],2), 'dev', round(ip['t_device_s'],2), round(d.get('positions_per_sec_device_only')))
    except Exception as e: print(f, 'ERR', e)
PY
for f in gpurun_out/r4b/c3_p*.err; do tail -n 2 $f; done
