# "z" / "b" streams end to end: compression on its own stream (default) against behind the assembly on the compute stream (GDBAMD_BGZF_STREAM=0), alternating;
# then the text kernel on the shapes of BASELINE configs[2] and [4] (10 000 samples; long PL columns)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6zs; mkdir -p $o
leg() {
python - "$o/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    z = d["stream_end_to_end_bgzf"]; b = d["stream_end_to_end_bcf"]["b"]
    print("%-12s z %.2f M pos/s (drain %.3f s, ratio %.2f)   b %.2f M pos/s (drain %.3f s, ratio %.2f)" % (sys.argv[2], z["positions_per_sec"] / 1e6, z["t_drain_s"], z["compression_ratio"], b["positions_per_sec"] / 1e6, b["t_drain_s"], b["compression_ratio"]))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
for i in 1 2; do for m in 1 0; do
  GDBAMD_BGZF_STREAM=$m python bench.py --steps 2 --warmup 1 --lanes 1 --no-c3 --no-cpu-baseline > $o/stream${m}_$i.json 2> $o/stream${m}_$i.err; leg stream${m}_$i
done; done 2>&1 | tee $o/result.txt
timeout 600 python -m pytest tests/test_bgzf.py tests/test_vcf_index.py -m gpu -x -q 2>&1 | tail -2 | tee -a $o/result.txt
for t in 1 0; do
  echo "text kernel = $t, 10 000 samples x 20 kb" | tee -a $o/result.txt
  GDBAMD_BGZF_TEXT=$t timeout 600 python tests/tools/bgzf_bench.py 10000 20000 z 2>&1 | grep "^format" | tee -a $o/result.txt
done
