cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
python bench.py > gpurun_out/r2p/bench_line.json 2> gpurun_out/r2p/bench.err
cat gpurun_out/r2p/bench_line.json | cut -c1-600
bash tests/tools/prof_stats.sh r2p/stats --no-stream > gpurun_out/r2p/stats.log 2>&1; head -12 gpurun_out/r2p/stats.log
bash tests/tools/prof_traffic.sh r2p/traffic --steps 2 --warmup 1 --no-stream > gpurun_out/r2p/traffic.log 2>&1
python bench.py --bcf --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2p/bench_bcf.json 2>/dev/null; cut -c1-400 gpurun_out/r2p/bench_bcf.json
python tests/tools/c5_sanity.py 10000 2000 2>&1 | tail -2 > gpurun_out/r2p/c5.txt; cat gpurun_out/r2p/c5.txt
./tests/tools/microbench/store_bw 32 > gpurun_out/r2p/store_bw.txt 2>&1
