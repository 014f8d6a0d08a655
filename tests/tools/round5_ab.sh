# round-5 A/B measurements, each alternating inside ONE gpurun call (boxes differ by ~5 %): gpurun -- 'bash tests/tools/round5_ab.sh <what>'
#   compact  resolved matrix 5 B (GDBAMD_RES_COMPACT=1) against 8 B          -> profiles/r5_ab_compact_matrix.txt
#   image    page image 8 / 4 / 6 KiB                                        -> profiles/r5_ab_image.txt
#   lanes    1 / 2 / 3 windows in flight (bench.py --lanes)                  -> profiles/r5_ab_lanes.txt
#   knobs    run lengths, image, order block, 2 lanes under 3 lanes          -> profiles/r5_knob_sweep_3lanes.txt
#   c5       tests/tools/c5_full.py with 1,3,2,1 pieces in flight            -> profiles/r5_c5_full_50000x100kb_lanes.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5ab_$1; mkdir -p $o
line() { # name: prints ms/step, positions/s, phases and the page kernel's launch time of $o/$1.json
python - "$o/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %.2f ms/step  %.2f M pos/s  %s  page kernel %.2f ms" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, {k: round(v, 2) for k, v in d["phase_ms"].items()}, d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
}
run() { name=$1; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass > $o/$name.json 2> $o/$name.err; line $name; }
case "$1" in
  compact) for i in 1 2 3; do run compact1_$i GDBAMD_RES_COMPACT=1 GDBAMD_BENCH_LANES=1; run compact0_$i GDBAMD_RES_COMPACT=0 GDBAMD_BENCH_LANES=1; done ;;
  image)   for i in 1 2 3; do for kb in 8 4 6; do run image${kb}_$i GDBAMD_WRITE_IMAGE_KB=$kb GDBAMD_BENCH_LANES=1; done; done ;;
  lanes)   for i in 1 2; do for l in 1 2 3; do run lanes${l}_$i GDBAMD_BENCH_LANES=$l; done; done ;;
  knobs)   for i in 1 2; do run base_$i A=1; run runw24_$i GDBAMD_RUN_W=24; run runw48_$i GDBAMD_RUN_W=48; run run64_$i GDBAMD_RUN=64; run run256_$i GDBAMD_RUN=256
             run img8_$i GDBAMD_WRITE_IMAGE_KB=8; run ob10_$i GDBAMD_ORDER_BLOCK_LOG2=10; run ob14_$i GDBAMD_ORDER_BLOCK_LOG2=14; run lanes2_$i GDBAMD_BENCH_LANES=2; done ;;
  c5)      C5_LANES=1,3,2,1 timeout 1200 python tests/tools/c5_full.py > $o/c5_lanes.json 2> $o/c5_lanes.err; grep "^{" $o/c5_lanes.json | cut -c1-700 ;;
  *) echo "usage: round5_ab.sh compact|image|lanes|knobs|c5" ;;
esac 2>&1 | tee $o/result.txt
