# round-6 first probe (one gpurun call): the default bench line on a fresh box, SQ counters of the non-page kernels, record types / slots of a c2 window
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r6_00; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench.err
tail -c 600 $o/bench.err
python tests/tools/type_stats.py 1000 1000000 > $o/type_stats.txt 2>&1; tail -3 $o/type_stats.txt
bash tests/tools/prof_sq.sh r6_00/sq "k_slots_light|k_assemble_size3|k_site_size|k_cell_ranges|k_assemble_write" --steps 3 --warmup 1 --lanes 1 --no-c3 > $o/sq.txt 2>&1
cat $o/sq.txt
