cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash tests/tools/prof_stats.sh wstats --steps 2 --warmup 1 --no-stream 2>&1 | grep -E "k_walk|k_cells" | head -8
