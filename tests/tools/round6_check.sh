# full GPU suite + the default bench line on the box (one gpurun call)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_10; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err; tail -c 300 $o/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_10/bench_line.json").read().strip().splitlines()[-1])
print("value %.2f M  ms/step %.2f  roofline %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"]))
for k in ("stream_end_to_end","stream_end_to_end_bgzf","stream_end_to_end_bcf","stream_end_to_end_bcf_bgzf","c3_streamed"):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if kk in ("positions_per_sec","GBps","ratio","compression_ratio","value","device_only_positions_per_sec","compress_GBps_of_input")})
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
