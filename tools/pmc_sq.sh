# SQ counter pass over the look-ahead sequence (tools/variant_bench.py); results -> gpurun_out/pmc_sq/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/a -- python3 $R/tools/variant_bench.py sq > $R/gpurun_out/pmc_sq_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/c -- python3 $R/tools/variant_bench.py sq2 > $R/gpurun_out/pmc_sq_c.log 2>&1
