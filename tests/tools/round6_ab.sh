# round-6 A/B measurements, each alternating inside ONE gpurun call (boxes differ by ~5 %): gpurun -- 'bash tests/tools/round6_ab.sh <what>'
#   warm     k_slots_light with / without the up-front column requests (GDBAMD_SLOT_WARM)   -> profiles/r6_ab_slot_warm.txt
#   store    k_slots_light: the workgroup stores its slots together / every lane its own / none -> profiles/r6_ab_slot_store.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6ab_$1; mkdir -p $o
line() {
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
  warm)    for i in 1 2 3; do run warm1_l1_$i GDBAMD_SLOT_WARM=1 GDBAMD_BENCH_LANES=1; run warm0_l1_$i GDBAMD_SLOT_WARM=0 GDBAMD_BENCH_LANES=1; done
           for i in 1 2; do run warm1_l3_$i GDBAMD_SLOT_WARM=1; run warm0_l3_$i GDBAMD_SLOT_WARM=0; done ;;
  store)   for i in 1 2 3; do run coop_l1_$i GDBAMD_SLOT_STORE=1 GDBAMD_BENCH_LANES=1; run lane_l1_$i GDBAMD_SLOT_STORE=0 GDBAMD_BENCH_LANES=1; done
           run none_l1 GDBAMD_SLOT_STORE=2 GDBAMD_BENCH_LANES=1
           for i in 1 2; do run coop_l3_$i GDBAMD_SLOT_STORE=1; run lane_l3_$i GDBAMD_SLOT_STORE=0; done ;;
  slotdbg) for i in 1 2; do run base_$i GDBAMD_BENCH_LANES=1; run onecell_$i GDBAMD_SLOT_DBG=8 GDBAMD_BENCH_LANES=1; run noemit_$i GDBAMD_SLOT_DBG=16 GDBAMD_BENCH_LANES=1; done ;;
  variant) # $2 = name under build/variants (the build before a change): the in-tree library against it
           for i in 1 2 3; do run new_l1_$i GDBAMD_BENCH_LANES=1; run old_l1_$i GDBAMD_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/$2/libgenomicsdb_amd.so GDBAMD_BENCH_LANES=1; done
           for i in 1 2; do run new_l3_$i A=1; run old_l3_$i GDBAMD_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/$2/libgenomicsdb_amd.so; done ;;
  lanes4)  for i in 1 2; do run l3_full_$i GDBAMD_BENCH_LANES=3; run l4_half_$i GDBAMD_BENCH_LANES=4 GDBAMD_BENCH_LANE_ARENA_MB=23040; run l3_half_$i GDBAMD_BENCH_LANES=3 GDBAMD_BENCH_LANE_ARENA_MB=23040; run l4_third_$i GDBAMD_BENCH_LANES=4 GDBAMD_BENCH_LANE_ARENA_MB=15360; done ;;
  micro)   # compile-time shapes (variant builds): the sizing pass at 64 registers, k_slots_light at 192 threads x 5 wavefronts per SIMD, the wide-strip k_slots_light at 192 threads (10 000 samples)
           V=$GRAFT_REPO_ROOT/build/variants
           for i in 1 2 3; do run base_$i GDBAMD_BENCH_LANES=1; run size3w8_$i GDBAMD_LIB_PATH=$V/size3w8/libgenomicsdb_amd.so GDBAMD_BENCH_LANES=1; run light192w5_$i GDBAMD_LIB_PATH=$V/light192w5/libgenomicsdb_amd.so GDBAMD_BENCH_LANES=1; done
           c3() { name=$1; shift; env "$@" python bench.py --stream-input --stream-source memory --samples 10000 --interval-bp 3000000 --window-bp 50000 --no-cpu-baseline > $o/$name.json 2> $o/$name.err
                  python - "$o/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-14s 10 000 samples x 3 Mb: %.3f M pos/s end to end, %.3f device-only" % (sys.argv[2], d["value"] / 1e6, d["positions_per_sec_device_only"] / 1e6))
PY
           }
           for i in 1 2; do c3 c3_base_$i A=1; c3 c3_wide192_$i GDBAMD_LIB_PATH=$V/wide192/libgenomicsdb_amd.so; done ;;
  *) echo "usage: round6_ab.sh warm|store|slotdbg|variant NAME|lanes4|micro" ;;
esac 2>&1 | tee $o/result.txt
