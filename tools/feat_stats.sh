#!/bin/bash
# per-kernel times of the look-ahead call under rocprofv3 --stats, image-side kernels first: tools/feat_stats.sh TAG [extra bench args]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r5/stats_$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --startup 0 --no-cpu-baseline --no-verify --no-per-frame --no-alone "$@" > $O.log 2>&1
python3 - <<P
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
img=("k_front","k_feat","k_filters","k_nms","k_scan_cells","k_emit","k_bin","k_halve","k_ingest","k_feat_order")
tot=0
for r in rows:
    n=r["Name"].replace("void ","")
    if n.startswith(img):
        tot+=float(r["AverageNs"])/1e3
        print("$TAG", n[:40].ljust(40), r["Calls"].rjust(5), "%9.1f us  min %7.1f max %7.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
print("$TAG image side, sum of averages per launch: %.1f us" % tot)
for r in rows[:8]:
    n=r["Name"].replace("void ","")
    print("$TAG   top:", n[:40].ljust(40), r["Calls"].rjust(5), "%9.1f us %5s %%" % (float(r["AverageNs"])/1e3, r["Percentage"]))
P
tail -c 400 $O.log | grep -o '"value": [0-9.]*' | head -1
