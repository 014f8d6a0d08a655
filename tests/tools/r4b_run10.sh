cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
C3="python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 10000000 --window-bp 50000 --arena-mb 49152"
run() { name=$1; shift; env "$@" $C3 > gpurun_out/r4b/c3x_$name.json 2> gpurun_out/r4b/c3x_$name.err; }
run warm A=1
run default A=1
run sub2048 GDBAMD_STAGE_SUB_MB=2048
run budget16g GDBAMD_STAGE_BUDGET_MB=16384 GDBAMD_STAGE_SUB_MB=2048
run nooverlap GDBAMD_OVERLAP_STAGING=0
run default2 A=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/c3x_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ip=d['input_path']
        print(f.split('c3x_')[1], round(d['value']), 'wall', round(ip['wall_s'],2), 'stage', round(ip['t_stage_s'],2), 'dev', round(ip['t_device_s'],2), 'gap', round(ip['wall_s']-ip['t_stage_s']-ip['t_device_s'],2), round(d.get('positions_per_sec_device_only')))
    except Exception as e: print(f, 'ERR', e)
PY
