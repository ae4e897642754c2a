#!/bin/bash
# variant of libvisomatch.so with vsm_dc.hip rebuilt under extra flags: tools/build_variant_dc.sh NAME "-DDC2_PHASE_TIMING"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/opencl-structure-from-motion_amd/csrc
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$1
FLAGS="-g -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include -I$C -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $2 -x hip -c $C/vsm_dc.hip -o $OUT/obj_$1/dc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libvisomatch_$1.so $C/build/vsm_kernels.o $C/build/vsm_api.o $C/build/vsm_host.o $C/build/vsm_ego.o $C/build/vsm_mono.o $OUT/obj_$1/dc.o
