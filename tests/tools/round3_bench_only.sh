# the bench line and its rocprofv3 evidence alone on a fresh box (tests/tools/round3_profiles.sh runs everything else).
# The default bench ends with ~20 s of sustained load (c3 leg, CPU baseline), after which the box's clocks stay lower for a while and
# k_assemble_write measures 3-9 % slower: the box idles before the profiler passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r3p; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err
cut -c1-400 $o/bench_line.json
sleep 120
bash tests/tools/prof_stats.sh r3p/stats --no-stream --no-c3 > $o/stats.log 2>&1; head -6 $o/stats.log
sleep 30
bash tests/tools/prof_traffic.sh r3p/traffic --steps 2 --warmup 1 --no-stream --no-c3 > $o/traffic.log 2>&1
python bench.py --bcf --steps 5 --warmup 1 --no-cpu-baseline --no-c3 > $o/bench_bcf.json 2>/dev/null; cut -c1-200 $o/bench_bcf.json
