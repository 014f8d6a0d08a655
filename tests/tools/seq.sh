cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "bcf or jni" 2>&1 | tail -5
python bench.py --bcf --steps 5 --warmup 1 --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bcf',d['value'], d['ms_per_step'], d['phase_ms'])"
bash tests/tools/prof_stats.sh bcfstats --bcf --steps 3 --warmup 1 --no-stream 2>&1 | head -8
