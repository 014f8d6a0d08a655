cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5e; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes" > $o/tests_lanes.log 2>&1; tail -3 $o/tests_lanes.log
for i in 1 2; do
  for l in 1 2 3; do
    python bench.py --steps 12 --warmup 3 --lanes $l --no-cpu-baseline --no-c3 --no-stream > $o/b_l${l}_$i.json 2> $o/b_l${l}_$i.err || tail -3 $o/b_l${l}_$i.err
    python - <<PY
import json
try:
    d=json.loads(open("$o/b_l${l}_$i.json").read().strip().splitlines()[-1])
    print("lanes=$l run $i: %.2f ms/step  %.2f M pos/s  %s  write kernel %.2f ms (frac %.3f)" % (d["ms_per_step"], d["value"]/1e6, {k: round(v, 2) for k, v in d["phase_ms"].items()}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("lanes=$l run $i failed:", e)
PY
  done
done 2>&1 | tee $o/ab_lanes.txt
