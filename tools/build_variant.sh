#!/bin/bash
# build a kernel-experiment variant of libvisomatch.so: tools/build_variant.sh NAME "-DVSM_UVL=1 -DVSM_MATCH_WAVES=5"
# -> gpurun_variants/libvisomatch_NAME.so (select at run time with VSM_LIB_PATH)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/opencl-structure-from-motion_amd/csrc
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$1
FLAGS="-g -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include -I$C -Wall -Wno-unused-result"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $2 -x hip -c $C/vsm_kernels.hip -o $OUT/obj_$1/k.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "Function Name: _Z7k_matchILi[24]ELb" | grep "Name\|VGPRs:\|ScratchSize\|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - -
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libvisomatch_$1.so $OUT/obj_$1/k.o $C/build/vsm_api.o $C/build/vsm_host.o $C/build/vsm_ego.o $C/build/vsm_mono.o $C/build/vsm_dc.o
