# round-5 first GPU call: the new parity tests, then the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5a; mkdir -p $o
timeout 1500 python -m pytest tests/test_reference_semantics.py tests/test_id_union_order.py -m gpu -x -q > $o/new_tests.log 2>&1; tail -3 $o/new_tests.log
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interior_window" > $o/spot_tests.log 2>&1; tail -5 $o/spot_tests.log
python bench.py > $o/bench_line.json 2> $o/bench.err
cut -c1-600 $o/bench_line.json
