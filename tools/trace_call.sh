#!/bin/bash
# rocprofv3 kernel trace of the look-ahead call (optionally under a library variant) + its per-stream timeline (tools/timeline.py)
# and the library's own per-chunk time line:  tools/trace_call.sh NAME [VARIANT]   -> gpurun_out/$ROUND/NAME.{log,timeline.txt,dbg.txt}
set -e
N=${1:-trace}
ROUND=${ROUND:-r4}
[ -n "$2" ] && [ "$2" != default ] && export VSM_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libvisomatch_$2.so
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$ROUND/$N
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps 2 --warmup 2 > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 2 > $OUT.timeline.txt 2>&1
tail -1 $OUT.timeline.txt
cd $GRAFT_REPO_ROOT && python3 tools/seq_debug_timing.py > $OUT.dbg.txt 2>&1
tail -6 $OUT.dbg.txt
rm -rf $OUT
