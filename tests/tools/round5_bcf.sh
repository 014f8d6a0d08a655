# BCF2 path: the parity tests that cover it, then the bench (1 and 3 windows in flight) and the kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5t; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q -k "bcf or lanes or format" 2>&1 | tail -3 | tee $o/tests_bcf.txt
for l in 1 3; do
  python bench.py --bcf --lanes $l --steps 6 --warmup 2 --no-cpu-baseline --no-c3 --no-stream > $o/bcf_l$l.json 2> $o/bcf_l$l.err
  python - "$o/bcf_l$l.json" "lanes=$l" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s %.2f ms/step  %.2f M pos/s  k_bcf_write %.2f ms (frac %.3f)  phases %s" % (sys.argv[2], d["ms_per_step"], d["value"] / 1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("phase_ms")))
PY
done 2>&1 | tee $o/bcf_bench.txt
bash tests/tools/prof_stats.sh r5t/stats_bcf --bcf --lanes 1 --steps 5 --warmup 1 --no-stream --no-c3 > $o/stats_bcf.log 2>&1; head -8 $o/stats_bcf.log
if [ -f build/variants/bcfprof/libgenomicsdb_amd.so ]; then GDBAMD_LIB_PATH=$PWD/build/variants/bcfprof/libgenomicsdb_amd.so python bench.py --bcf --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-c3 --no-stream --no-alone-pass 2>&1 | grep "k_bcf_write cycles" | tail -1 | tee $o/bcf_sections.txt; fi
