#!/bin/bash
# correctness of the device Delaunay chain + its phase times: tools/dc2_quick.sh TAG  (needs gpurun_variants/libvisomatch_phase.so)
mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/dc2_check.py > gpurun_out/r3/check_$1.log 2>&1; tail -1 gpurun_out/r3/check_$1.log
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t_$1.log 2>&1; tail -1 gpurun_out/r3/t_$1.log
VSM_LIB_PATH=$PWD/gpurun_variants/libvisomatch_phase.so timeout -k 10 120 python tools/dc2_phases.py 7400 76 2>&1 | grep -v "init\|write-out\|L5" > gpurun_out/r3/phase_$1.log; cat gpurun_out/r3/phase_$1.log
