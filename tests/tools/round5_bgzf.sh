# BGZF: two wavefronts per block (default) against one (GDBAMD_BGZF_WAVES=1), alternating; then the tests that inflate the streams
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5z; mkdir -p $o
timeout 900 python -m pytest tests/test_bgzf.py -x -q 2>&1 | tail -3
for i in 1 2; do for w in 2 1; do
  echo "waves per block = $w (run $i)" | tee -a $o/bgzf_ab.txt
  GDBAMD_BGZF_WAVES=$w timeout 300 python tests/tools/bgzf_bench.py 1000 200000 z,b 2>&1 | grep "^format" | tee -a $o/bgzf_ab.txt
done; done
timeout 900 python -m pytest tests/test_vcf_index.py tests/test_gpu_parity.py -m gpu -x -q -k "bgzf or index or tbi or z_ or golden" 2>&1 | tail -3
timeout 900 python tests/tools/fuzz.py 60 9000 2>&1 | tail -2
