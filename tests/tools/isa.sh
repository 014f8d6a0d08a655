#!/bin/bash
# device ISA of the pipeline (gfx950) as text: tests/tools/isa.sh [out.s]
OUT=${1:-/tmp/gdb_pipeline.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S "$(dirname "$0")/../../genomicsdb_amd/csrc/kernels/gdb_pipeline.hip" -o "$OUT" && echo "$OUT"
