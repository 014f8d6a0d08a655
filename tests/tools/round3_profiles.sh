# round-3 measurement set (GPU box): gpurun -- 'bash tests/tools/round3_profiles.sh'; summaries go to profiles/ via sync_profiles3.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r3p; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > $o/smoke.log 2>&1; tail -2 $o/smoke.log
python bench.py > $o/bench_line.json 2> $o/bench.err
cut -c1-700 $o/bench_line.json
bash tests/tools/prof_stats.sh r3p/stats --no-stream --no-c3 > $o/stats.log 2>&1; head -14 $o/stats.log
bash tests/tools/prof_traffic.sh r3p/traffic --steps 2 --warmup 1 --no-stream --no-c3 > $o/traffic.log 2>&1
python bench.py --bcf --steps 5 --warmup 1 --no-cpu-baseline --no-c3 > $o/bench_bcf.json 2>/dev/null; cut -c1-400 $o/bench_bcf.json
python tests/tools/bgzf_bench.py 1000 1000000 > $o/bgzf_bench.txt 2>&1; cat $o/bgzf_bench.txt
python tests/tools/type_stats.py > $o/type_stats.txt 2>&1; tail -5 $o/type_stats.txt
bash tests/tools/prof_sq_any.sh r3p/bgzf_sq k_bgzf_deflate python /root/repo/tests/tools/bgzf_bench.py 1000 400000 z 2>&1 | grep -v amdgpu.ids > $o/bgzf_sq_counters.txt; cat $o/bgzf_sq_counters.txt
python tests/tools/frag_bench.py 1000 1000000 2>&1 | grep -v "rocprofv3\|HSA version\|amdgpu.ids" > $o/frag_bench.txt; cat $o/frag_bench.txt
