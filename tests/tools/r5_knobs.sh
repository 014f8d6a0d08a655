cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5h; mkdir -p $o
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass > $o/$name.json 2> $o/$name.err || tail -2 $o/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$o/$name.json").read().strip().splitlines()[-1])
    print("%-28s %.2f ms/step  %.2f M pos/s" % ("$name", d["ms_per_step"], d["value"]/1e6))
except Exception as e: print("$name failed", e)
PY
}
for rep in 1 2; do
run base_$rep A=1
run runw24_$rep GDBAMD_RUN_W=24
run runw48_$rep GDBAMD_RUN_W=48
run run64_$rep GDBAMD_RUN=64
run run256_$rep GDBAMD_RUN=256
run img8_$rep GDBAMD_WRITE_IMAGE_KB=8
run ob10_$rep GDBAMD_ORDER_BLOCK_LOG2=10
run ob14_$rep GDBAMD_ORDER_BLOCK_LOG2=14
run lanes2_$rep GDBAMD_BENCH_LANES=2
done 2>&1 | tee $o/knobs.txt
