# round-6 measurement set (GPU box): gpurun -- 'bash tests/tools/round6_profiles.sh [notests]'; summaries go to profiles/ via sync_profiles6.sh
# Order as in round 4: the default bench line first on the fresh box, two idle minutes, then the profiler passes, then everything else.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6p; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err
cut -c1-400 $o/bench_line.json
python bench.py --lanes 1 --no-stream --no-c3 --no-cpu-baseline > $o/bench_line_lanes1.json 2> $o/bench_lanes1.err
cut -c1-300 $o/bench_line_lanes1.json
sleep 120
bash tests/tools/prof_stats.sh r6p/stats --no-stream --no-c3 > $o/stats.log 2>&1; head -6 $o/stats.log
bash tests/tools/prof_stats.sh r6p/stats_lanes1 --lanes 1 --no-stream --no-c3 > $o/stats_lanes1.log 2>&1; head -4 $o/stats_lanes1.log
sleep 30
bash tests/tools/prof_traffic.sh r6p/traffic --steps 2 --warmup 1 --lanes 1 --no-stream --no-c3 > $o/traffic.log 2>&1
bash tests/tools/prof_stats.sh r6p/stats_bcf --bcf --lanes 1 --steps 5 --warmup 1 --no-stream --no-c3 > $o/stats_bcf.log 2>&1; head -4 $o/stats_bcf.log
bash tests/tools/prof_traffic.sh r6p/traffic_bcf --bcf --steps 2 --warmup 1 --lanes 1 --no-stream --no-c3 --no-alone-pass > $o/traffic_bcf.log 2>&1
python bench.py --bcf --steps 6 --warmup 3 --no-cpu-baseline --no-c3 > $o/bench_bcf.json 2>/dev/null; cut -c1-300 $o/bench_bcf.json
python bench.py --bcf --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline --no-c3 --no-stream > $o/bench_bcf_lanes1.json 2>/dev/null; cut -c1-300 $o/bench_bcf_lanes1.json
bash tests/tools/prof_stats.sh r6p/stats_z --steps 2 --warmup 1 --lanes 1 --no-c3 > $o/stats_z.log 2>&1; head -4 $o/stats_z.log      # (with the stream legs: the BGZF kernels of the "z" / "b" streams)
for t in 1 0; do echo "text kernel = $t, 1 000 samples x 200 kb"; GDBAMD_BGZF_TEXT=$t timeout 300 python tests/tools/bgzf_bench.py 1000 200000 z,b 2>&1 | grep "^format"; done > $o/bgzf_ab.txt 2>&1
for t in 1 0; do echo "text kernel = $t, 10 000 samples x 20 kb"; GDBAMD_BGZF_TEXT=$t timeout 600 python tests/tools/bgzf_bench.py 10000 20000 z 2>&1 | grep "^format"; done >> $o/bgzf_ab.txt 2>&1
cat $o/bgzf_ab.txt
C5_LANES=1,2 timeout 900 python tests/tools/c5_full.py > $o/c5_full.json 2> $o/c5_full.err; cut -c1-400 $o/c5_full.json
timeout 300 python __graft_entry__.py smoke > $o/smoke.log 2>&1; tail -2 $o/smoke.log
if [ "$1" != "notests" ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -n "passed\|failed" $o/gpu_tests.log
  GDBAMD_SIZE3_CHECK=1 timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests_size3_check.log 2>&1; grep -n "passed\|failed" $o/gpu_tests_size3_check.log
fi
