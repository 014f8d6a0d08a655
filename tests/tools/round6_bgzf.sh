# BGZF: the anchored text kernel (default for "z") against the byte-level one (GDBAMD_BGZF_TEXT=0), alternating, on the shapes of BASELINE
# configs[1] (1 000 samples), [2] (10 000 samples) and [4] (long PL columns); before that the tests that inflate the streams
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6z; mkdir -p $o; rm -f $o/bgzf_ab.txt
timeout 900 python -m pytest tests/test_bgzf.py tests/test_vcf_index.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do for t in 1 0; do
  echo "text kernel = $t (run $i), 1 000 samples x 200 kb" | tee -a $o/bgzf_ab.txt
  GDBAMD_BGZF_TEXT=$t timeout 300 python tests/tools/bgzf_bench.py 1000 200000 z 2>&1 | grep "^format" | tee -a $o/bgzf_ab.txt
done; done
for t in 1 0; do
  echo "text kernel = $t, 10 000 samples x 20 kb" | tee -a $o/bgzf_ab.txt
  GDBAMD_BGZF_TEXT=$t timeout 600 python tests/tools/bgzf_bench.py 10000 20000 z 2>&1 | grep "^format" | tee -a $o/bgzf_ab.txt
done
