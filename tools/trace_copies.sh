#!/bin/bash
# kernel + memory-copy trace of the look-ahead call (inputs resident in HBM): one merged time line of the last call - every
# copy (direction, bytes) and every kernel with its stream / agent - to see who waits for the DMA engines.
#   [VSM_PY_OPTIONS=...] tools/trace_copies.sh NAME  -> gpurun_out/$ROUND/NAME.merged.txt
N=${1:-copies}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r4}/$N
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/tools/seq_debug_timing.py > $OUT.log 2>&1
python3 - > $OUT.merged.txt <<PY
import csv, glob, re
k = sorted(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
c = sorted(csv.DictReader(open(glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(k) if r["Kernel_Name"].startswith("k_front")]
i0 = fr[-3]  # the last call's first k_front (three chunks per call)
t0 = int(k[i0]["Start_Timestamp"])
ev = []
for r in k[i0:]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "s%-2s %-26s grid %sx%s" % (r["Stream_Id"], n[:26], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"])))
for r in c:
    if int(r["Start_Timestamp"]) >= t0:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "s%-2s COPY %-18s %s" % (r.get("Stream_Id", "?"), r["Direction"].replace("MEMORY_COPY_", ""), " ".join("%s=%s" % (a, r[a]) for a in r if a in ("Bytes", "Size"))) ))
for s, e, what in sorted(ev):
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, what))
PY
tail -8 $OUT.log | grep -v rocprof
rm -rf $OUT
