cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5j; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_one_piece or high_alt or c5_50000 or c3_width_10000 or overflow" > $o/tests.log 2>&1; tail -25 $o/tests.log | cut -c1-220
C5_LANES=1,3,2,1 timeout 1200 python tests/tools/c5_full.py > $o/c5_lanes.json 2> $o/c5_lanes.err; tail -3 $o/c5_lanes.err
python - <<PY
import json
for l in open("$o/c5_lanes.json"):
    if l.startswith("{"):
        d=json.loads(l)
        print("c5 lanes=%d: wall %.3f s device-sum %.3f s  %s  %.0f GB/s" % (d["lanes"], d["wall_s"], d["device_s"], {k: round(v) for k,v in d["phase_ms"].items()}, d["GBps_out"]))
PY
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-c3 --no-stream > $o/b.json 2> $o/b.err
python - <<PY
import json
d=json.loads(open("$o/b.json").read().strip().splitlines()[-1])
print("c2 default: %.2f ms/step %.2f M pos/s; alone %s" % (d["ms_per_step"], d["value"]/1e6, d["roofline"].get("alone",{}).get("avg_launch_ms")))
PY
