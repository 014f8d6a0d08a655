cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "bcf or jni or allele_specific" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-stream --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('text', d['roofline']['avg_launch_ms'])"
python bench.py --bcf --steps 3 --warmup 1 --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bcf',d['value'], d['ms_per_step'], d['phase_ms'])"
