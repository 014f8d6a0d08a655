cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
C3="python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 6000000 --window-bp 50000 --arena-mb 49152"
run() { name=$1; shift; env "$@" $C3 > gpurun_out/r4b/c3y_$name.json 2> gpurun_out/r4b/c3y_$name.err; }
run warm A=1
run s8_a GDBAMD_SIZE3=8
run s0_a GDBAMD_SIZE3=0
run s8_b GDBAMD_SIZE3=8
run s0_b GDBAMD_SIZE3=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/c3y_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ip=d['input_path']
        print(f.split('c3y_')[1], round(d['value']), 'wall', round(ip['wall_s'],2), 'stage', round(ip['t_stage_s'],2), 'dev', round(ip['t_device_s'],2), round(d.get('positions_per_sec_device_only')))
    except Exception as e: print(f, 'ERR', e)
PY
GDBAMD_SIZE3=0 python tests/tools/c5_full.py 2>/dev/null | cut -c300-700
GDBAMD_SIZE3=8 python tests/tools/c5_full.py 2>/dev/null | cut -c300-700
