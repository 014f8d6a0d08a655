# the round's last sources: default bench line, BGZF kernels on both shapes, smoke, the GPU suite (one gpurun call)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6f; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err; cut -c1-300 $o/bench_line.json
for t in 1 0; do echo "text kernel = $t, 1 000 samples x 200 kb"; GDBAMD_BGZF_TEXT=$t timeout 300 python tests/tools/bgzf_bench.py 1000 200000 z,b 2>&1 | grep "^format"; done > $o/bgzf_ab.txt 2>&1
for t in 1 0; do echo "text kernel = $t, 10 000 samples x 20 kb"; GDBAMD_BGZF_TEXT=$t timeout 600 python tests/tools/bgzf_bench.py 10000 20000 z 2>&1 | grep "^format"; done >> $o/bgzf_ab.txt 2>&1
cat $o/bgzf_ab.txt
timeout 300 python __graft_entry__.py smoke > $o/smoke.log 2>&1; tail -1 $o/smoke.log
timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -n "passed\|failed" $o/gpu_tests.log
