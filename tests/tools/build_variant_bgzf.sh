#!/bin/bash
# A/B builds of the BGZF kernels with extra -D flags: tests/tools/build_variant_bgzf.sh NAME "flags" -> build/variants/NAME/libgenomicsdb_amd.so
set -e
cd "$(dirname "$0")/../.."
d=build/variants/$1; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -gline-tables-only -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result $2 -x hip -c genomicsdb_amd/csrc/kernels/gdb_bgzf.hip -o $d/bgzf.o
objs=$(ls build/obj/*.o | grep -v kernels_gdb_bgzf)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgenomicsdb_amd.so $d/bgzf.o $objs -lz
echo built $d
