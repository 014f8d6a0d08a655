cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5c; mkdir -p $o
for kb in 8 6 4; do for e in 1 2 3; do
  echo "image ${kb} KB, engines $e:" | tee -a $o/overlap.txt
  GDBAMD_WRITE_IMAGE_KB=$kb ARENA_MB=46000 timeout 300 python tests/tools/overlap_probe.py $e 12 2>&1 | tail -1 | tee -a $o/overlap.txt
done; done
timeout 900 python -m pytest tests/test_gpu_asm_paths.py -x -q -k "wide_matrix" 2>&1 | tail -2
