# long runs of the round: c3 at its stated size (10 000 samples x all of chr1, one pass) and a longer differential fuzz
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5l; mkdir -p $o
timeout 1500 python bench.py --c3-full > $o/c3_full.json 2> $o/c3_full.err; tail -1 $o/c3_full.json | cut -c1-700
timeout 2400 python tests/tools/fuzz.py 600 20000 > $o/fuzz600.log 2>&1; tail -2 $o/fuzz600.log
