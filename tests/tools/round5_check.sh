cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5q; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_genome.py -m gpu -x -q -k "histogram" 2>&1 | tail -2
GDBAMD_SIZE3_CHECK=1 timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests_size3_check.log 2>&1; grep -n "passed\|failed" $o/gpu_tests_size3_check.log
timeout 1500 python tests/tools/fuzz.py 200 5000 > $o/fuzz.log 2>&1; tail -3 $o/fuzz.log
