#!/bin/bash
# kernel + memory-copy trace of the look-ahead call fed from host memory: when the frames arrive, when the features run
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r4}/hostfed
rm -rf $OUT; mkdir -p $OUT
SEQ_HOST_INPUTS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/tools/seq_debug_timing.py > $OUT.log 2>&1
python3 - <<PY
import csv, glob
k = sorted(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
c = sorted(csv.DictReader(open(glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
fr = [i for i, r in enumerate(k) if r["Kernel_Name"].startswith("k_front")]
i0 = fr[-3]  # the last call's first k_front
t_first = int(k[i0]["Start_Timestamp"])
# the call starts with its first H2D piece: the last group of H2D copies before that k_front
h2d = [r for r in c if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
last = [r for r in h2d if int(r["Start_Timestamp"]) > t_first - 3_000_000]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("H2D  %8.1f us  dur %7.1f" % ((s - t0) / 1e3, (e - s) / 1e3))
for r in k[i0:]:
    n = r["Kernel_Name"].split("(")[0]
    if n.startswith(("k_front", "k_match", "k_refine")):
        print("%-28s %8.1f us  dur %7.1f" % (n[:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $OUT
