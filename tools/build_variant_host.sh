#!/bin/bash
# variant of libvisomatch.so with vsm_host.cpp rebuilt under extra flags: tools/build_variant_host.sh NAME "-DVSM_SORT_WORD=0"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/opencl-structure-from-motion_amd/csrc
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$1
FLAGS="-g -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include -I$C -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS $2 -x c++ -c $C/vsm_host.cpp -o $OUT/obj_$1/host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libvisomatch_$1.so $C/build/vsm_kernels.o $C/build/vsm_api.o $OUT/obj_$1/host.o $C/build/vsm_ego.o $C/build/vsm_mono.o $C/build/vsm_dc.o
