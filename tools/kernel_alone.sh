#!/bin/bash
# per-kernel times of the device chain alone (tools/dc2_bench.py under rocprofv3 --stats): tools/kernel_alone.sh TAG [n] [copies]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/alone_$1
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $GRAFT_REPO_ROOT/tools/dc2_bench.py ${2:-7400} ${3:-67} > $O.log 2>&1
python3 - <<P
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print("$1", r["Name"][:44].ljust(44), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3))
P
