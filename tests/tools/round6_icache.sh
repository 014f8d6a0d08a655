# instruction-cache counters of the non-page kernels (one rocprofv3 --pmc pass per group): gpurun -- 'bash tests/tools/round6_icache.sh'
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out/r6_icache; mkdir -p $out
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"; do
  d=$out/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_slots_light|k_assemble_size3|k_site_size|k_cell_ranges|k_slots_heavy|k_site_copy" --output-format csv -d $d -o p -- python /root/repo/bench.py --steps 3 --warmup 1 --lanes 1 --no-c3 --no-cpu-baseline --no-stream </dev/null > $out/run$i.log 2>&1
  tail -2 $out/run$i.log | cut -c1-200
  i=$((i+1))
done
ls $out/*/
