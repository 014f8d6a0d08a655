cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r5d; mkdir -p $o
for i in 1 2 3; do
  for kb in 8 4 6; do
    GDBAMD_WRITE_IMAGE_KB=$kb python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c3 --no-stream > $o/b_${kb}_$i.json 2> $o/b_${kb}_$i.err
    python - <<PY
import json
d=json.loads(open("$o/b_${kb}_$i.json").read().strip().splitlines()[-1])
print("image=${kb}KB run $i: %.2f ms/step  %s  write %.2f ms" % (d["ms_per_step"], {k: round(v, 2) for k, v in d["phase_ms"].items()}, d["roofline"]["avg_launch_ms"]))
PY
  done
done 2>&1 | tee $o/ab_image.txt
