cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5g; mkdir -p $o
GDBAMD_STREAM_TRACE=1 python bench.py --steps 6 --warmup 3 --no-c3 --no-cpu-baseline > $o/b_trace.json 2> $o/b_trace.err
grep -v "^$" $o/b_trace.err | head -60
python - <<PY
import json
d=json.loads(open("$o/b_trace.json").read().strip().splitlines()[-1])
print("lanes3 trace run: %.2f ms/step; stream %s" % (d["ms_per_step"], {k: d["stream_end_to_end"][k] for k in ("positions_per_sec","t_first_byte_s","t_producing_s","t_drain_s")}))
PY
for l in 1 2 3; do
GDBAMD_BENCH_C3_LANES=$l python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 10000000 --window-bp 50000 > $o/c3_l$l.json 2> $o/c3_l$l.err || tail -3 $o/c3_l$l.err
python - <<PY
import json
try:
    d=json.loads(open("$o/c3_l$l.json").read().strip().splitlines()[-1])
    print("c3 lanes=$l: %.3f M pos/s wall %.2f s  %s" % (d["value"]/1e6, d["wall_accounting"]["wall_s"], {k: round(v,3) for k,v in d["wall_accounting"].items() if k.endswith("_s")}))
except Exception as e: print("c3 lanes=$l failed", e)
PY
done
