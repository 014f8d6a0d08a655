cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k pinned 2>&1 | tail -40 > gpurun_out/r4b/pinned_test.txt
cat gpurun_out/r4b/pinned_test.txt | head -60
bash tests/tools/round4_profiles.sh
