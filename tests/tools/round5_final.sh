# after the last source change: the tests that cover it, the default bench line and the PMC traffic passes of the final kernel sources
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5p; mkdir -p $o
timeout 900 python -m pytest tests/test_info_genotype_vectors.py tests/test_reference_semantics.py tests/test_gpu_asm_paths.py -m gpu -x -q 2>&1 | tail -2 | tee $o/final_tests.txt
python bench.py > $o/bench_line.json 2> $o/bench.err; cut -c1-300 $o/bench_line.json
bash tests/tools/prof_traffic.sh r5p/traffic --steps 2 --warmup 1 --lanes 1 --no-stream --no-c3 > $o/traffic.log 2>&1
bash tests/tools/prof_traffic.sh r5p/traffic_bcf --bcf --steps 2 --warmup 1 --lanes 1 --no-stream --no-c3 --no-alone-pass > $o/traffic_bcf.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -n "passed\|failed" $o/gpu_tests.log
