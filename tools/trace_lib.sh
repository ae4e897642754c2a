#!/bin/bash
# kernel trace of the look-ahead call under a library variant, lines of one kernel: tools/trace_lib.sh VARIANT PATTERN
[ "$1" != default ] && export VSM_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libvisomatch_$1.so
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/tl_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps 2 --warmup 2 > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 2 | grep "$2" | cut -c1-70 | sed "s/^/$1 /"
