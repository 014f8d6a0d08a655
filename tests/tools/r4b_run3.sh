mkdir -p gpurun_out/r4b
for cfg in "1 8" "2 8" "2 4" "2 6" "3 4"; do
set -- $cfg
echo "engines=$1 image_kb=$2" >> gpurun_out/r4b/overlap.txt
GDBAMD_WRITE_IMAGE_KB=$2 timeout 300 python tests/tools/overlap_probe.py $1 12 >> gpurun_out/r4b/overlap.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r4b/overlap.txt
