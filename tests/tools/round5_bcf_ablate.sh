# where k_bcf_write's time goes: the kernel with parts switched off (GDBAMD_BCF_ABLATE bits: 1 no flush stores, 2 no entry fetch, 4 no formatting, 8 no resolved-row loads); the output is wrong, only the duration counts
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5s; mkdir -p $o
for a in 0 1 2 4 6 7 8 14 15 0; do
  GDBAMD_BCF_ABLATE=$a python bench.py --bcf --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass > $o/a$a.json 2> $o/a$a.err
  python - "$o/a$a.json" "ablate=$a" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s %.2f ms/step  k_bcf_write %.2f ms  phases %s" % (sys.argv[2], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("phase_ms")))
PY
done 2>&1 | tee $o/ablate.txt
