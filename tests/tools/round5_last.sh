# the final sources once more: the GPU suite under the sizing cross-check and a fuzz run with other seeds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5z; mkdir -p $o
GDBAMD_SIZE3_CHECK=1 timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests_size3_check.log 2>&1; grep -n "passed\|failed" $o/gpu_tests_size3_check.log
timeout 1800 python tests/tools/fuzz.py 400 47000 > $o/fuzz.log 2>&1; tail -2 $o/fuzz.log
