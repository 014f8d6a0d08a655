cd $GRAFT_REPO_ROOT
GDBAMD_EVENTS=1 python -m pytest tests -m gpu -x -q -k "golden and stream" 2>&1 | tail -2
for d in 0; do GDBAMD_EV_DBG=$d GDBAMD_EVENTS=1 python bench.py --no-cpu-baseline --no-stream --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('events dbg=$d',d['value'], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['phase_ms'])"; done
python bench.py --no-cpu-baseline --no-stream --steps 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('matrix',d['value'], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['phase_ms'])"
