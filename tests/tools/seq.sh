cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('text',d['value'], d['roofline']['avg_launch_ms'], d['ms_per_step'], d['phase_ms'])"
