# c5 at its stated size: the cooperative long-entry copy with 4 words per lane in flight against word by word, then per-kernel stats of a fifth of the region
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5i; mkdir -p $o
timeout 900 python tests/tools/c5_full.py > $o/c5_u4.json 2> $o/c5_u4.err; tail -1 $o/c5_u4.json | cut -c1-900
GDBAMD_COOP_UNROLL=0 timeout 900 python tests/tools/c5_full.py > $o/c5_u1.json 2> $o/c5_u1.err; tail -1 $o/c5_u1.json | cut -c1-900
timeout 900 python tests/tools/c5_full.py > $o/c5_u4b.json 2> $o/c5_u4b.err; tail -1 $o/c5_u4b.json | cut -c1-900
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$o/prof -o p -- python /root/repo/tests/tools/c5_full.py 50000 20000 > /root/repo/$o/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then python3 tests/tools/short_stats.py "$f" | head -40 | tee $o/c5_kernel_stats_short.txt; cp "$f" $o/c5_kernel_stats.csv; fi
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_one_piece or high_alt or c5_50000" 2>&1 | tail -3
rm -rf $o/prof
