# round-4 measurement set (GPU box): gpurun -- 'bash tests/tools/round4_profiles.sh [notests]'; summaries go to profiles/ via sync_profiles4.sh
# Order: the bench line first on the fresh box, two idle minutes, then the profiler passes (a box measures k_assemble_write 3-9 % slower for
# a while after sustained load - profiles/README.md, round 3), then everything else.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r4p; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err
cut -c1-500 $o/bench_line.json
sleep 120
bash tests/tools/prof_stats.sh r4p/stats --no-stream --no-c3 > $o/stats.log 2>&1; head -6 $o/stats.log
sleep 30
bash tests/tools/prof_traffic.sh r4p/traffic --steps 2 --warmup 1 --no-stream --no-c3 > $o/traffic.log 2>&1
python bench.py --bcf --steps 5 --warmup 1 --no-cpu-baseline --no-c3 > $o/bench_bcf.json 2>/dev/null; cut -c1-300 $o/bench_bcf.json
timeout 300 python __graft_entry__.py smoke > $o/smoke.log 2>&1; tail -2 $o/smoke.log
if [ "$1" != "notests" ]; then timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log; fi
