# the page kernel on a high-priority stream (GDBAMD_PAGE_PRIORITY=1) against the lanes' own streams, 3 windows in flight; then the BCF2 kernels' stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5r; mkdir -p $o
for i in 1 2 3; do for pr in 0 1; do
  GDBAMD_PAGE_PRIORITY=$pr python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass > $o/p${pr}_$i.json 2> $o/p${pr}_$i.err
  python - "$o/p${pr}_$i.json" "priority=$pr run $i" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-20s %.2f ms/step  %.2f M pos/s  page kernel %.2f ms (frac %.3f)" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
done; done 2>&1 | tee $o/ab_priority.txt
bash tests/tools/prof_stats.sh r5r/stats_bcf --bcf --lanes 1 --steps 5 --warmup 1 --no-stream --no-c3 > $o/stats_bcf.log 2>&1; head -16 $o/stats_bcf.log
