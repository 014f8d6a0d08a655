# round 5: the resolved matrix compact (5 B per pair) against wide (8 B), alternating inside one call; then the tests that touch it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5b; mkdir -p $o
for i in 1 2 3; do
  for c in 1 0; do
    GDBAMD_RES_COMPACT=$c python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c3 --no-stream > $o/bench_c${c}_$i.json 2> $o/bench_c${c}_$i.err
    python - <<PY
import json
d=json.loads(open("$o/bench_c${c}_$i.json").read().strip().splitlines()[-1])
print("compact=$c run $i: %.2f ms/step  %s  write %.2f ms" % (d["ms_per_step"], {k: round(v, 2) for k, v in d["phase_ms"].items()}, d["roofline"]["avg_launch_ms"]))
PY
  done
done
timeout 1500 python -m pytest tests/test_gpu_asm_paths.py -x -q -k "compact or wide_matrix" > $o/tests_compact.log 2>&1; tail -3 $o/tests_compact.log
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interior_window or c2_scale or golden" > $o/tests_parity.log 2>&1; tail -3 $o/tests_parity.log
