#!/bin/bash
# rocprofv3 kernel trace of the look-ahead call + its per-stream timeline (tools/timeline.py): tools/trace_r3.sh NAME
set -e
N=${1:-trace}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/$N
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-per-frame --no-verify --no-alone --steps 2 --warmup 2 > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 4 > $OUT.timeline.txt 2>&1
tail -3 $OUT.timeline.txt
