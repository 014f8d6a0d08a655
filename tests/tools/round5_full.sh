# the whole GPU suite and a fuzz run on the sources as they are
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5x; mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log
timeout 1500 python tests/tools/fuzz.py 300 31000 > $o/fuzz.log 2>&1; tail -2 $o/fuzz.log
