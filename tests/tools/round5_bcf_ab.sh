# k_bcf_write: 64-byte against 96-byte entry slots (build/variants/cap96), alternating on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5w; mkdir -p $o
for i in 1 2 3; do for v in default cap96; do
  lib=""; [ $v = cap96 ] && lib=$PWD/build/variants/cap96/libgenomicsdb_amd.so
  GDBAMD_LIB_PATH=$lib python bench.py --bcf --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass > $o/$v$i.json 2> $o/$v$i.err
  python - "$o/$v$i.json" "$v run $i" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-16s %.2f ms/step  k_bcf_write %.2f ms" % (sys.argv[2], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done; done 2>&1 | tee $o/ab_cap.txt
