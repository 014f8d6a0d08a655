# BGZF: the anchored text kernel (default for "z") against the byte-level one (GDBAMD_BGZF_TEXT=0), alternating; then the tests that inflate the streams
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6z; mkdir -p $o
timeout 900 python -m pytest tests/test_bgzf.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do for t in 1 0; do
  echo "text kernel = $t (run $i)" | tee -a $o/bgzf_ab.txt
  GDBAMD_BGZF_TEXT=$t timeout 300 python tests/tools/bgzf_bench.py 1000 200000 z 2>&1 | grep "^format" | tee -a $o/bgzf_ab.txt
done; done
