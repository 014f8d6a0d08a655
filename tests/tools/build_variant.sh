#!/bin/bash
# A/B builds of the kernels file with extra -D flags: tests/tools/build_variant.sh NAME "flags" -> build/variants/NAME/libgenomicsdb_amd.so
# (run it with GDBAMD_LIB_PATH=... python bench.py; the other objects come from build/obj of the normal build)
set -e
cd "$(dirname "$0")/../.."
d=build/variants/$1; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -gline-tables-only -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $2 -x hip -c genomicsdb_amd/csrc/kernels/gdb_pipeline.hip -o $d/pipeline.o
objs=$(ls build/obj/*.o | grep -v kernels_gdb_pipeline)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgenomicsdb_amd.so $d/pipeline.o $objs -lz
echo built $d
