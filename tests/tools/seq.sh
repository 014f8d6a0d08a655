cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -12
mkdir -p gpurun_out/c3
GDBAMD_STREAM_TRACE=1 GDBAMD_STAGE_BUDGET_MB=2048 python bench.py --stream-input --samples 10000 --interval-bp 1000000 --window-bp 50000 --no-cpu-baseline > gpurun_out/c3/line.json 2> gpurun_out/c3/trace.txt
grep "gdbamd stage" gpurun_out/c3/trace.txt | sed -n 3,8p
python -c "
import json;d=json.load(open('gpurun_out/c3/line.json'));print(d['value'], d['input_path'], d.get('positions_per_sec_excluding_generator'))"
