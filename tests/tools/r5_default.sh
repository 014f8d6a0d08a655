cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5f; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err; tail -2 $o/bench.err
python - <<PY
import json
d=json.loads(open("$o/bench_line.json").read().strip().splitlines()[-1])
print("default: %.2f ms/step %.2f M pos/s lanes %s; roofline %s; alone %s" % (d["ms_per_step"], d["value"]/1e6, d.get("lanes"), {k: d["roofline"][k] for k in ("avg_launch_ms","frac")}, d["roofline"].get("alone")))
for k in ("stream_end_to_end","stream_end_to_end_bgzf","stream_end_to_end_bcf","c3_streamed"):
    v=d.get(k)
    if isinstance(v,dict): print(k, v.get("positions_per_sec", v.get("value")), v.get("wall_accounting"))
PY
for l in 4 1; do
python bench.py --lanes $l --steps 12 --warmup 4 --no-cpu-baseline --no-c3 --no-stream > $o/b_l$l.json 2> $o/b_l$l.err || tail -3 $o/b_l$l.err
python - <<PY
import json
try:
    d=json.loads(open("$o/b_l$l.json").read().strip().splitlines()[-1])
    print("lanes=$l: %.2f ms/step  %.2f M pos/s write kernel %.2f ms" % (d["ms_per_step"], d["value"]/1e6, d["roofline"]["avg_launch_ms"]))
except Exception as e: print("lanes=$l failed", e)
PY
done
timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; tail -4 $o/gpu_tests.log
