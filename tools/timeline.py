#!/usr/bin/env python3
"""Timeline of the LAST look-ahead call in a rocprofv3 kernel trace (kernel_trace.csv): one line per kernel launch with
its stream, start (µs after the call's first kernel), duration and grid, then per-stream busy time.
  python tools/timeline.py <kernel_trace.csv> [n_calls_back=1]"""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fronts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("k_front", "k_ingest"))]
# a call = one period of the front kernels' grid sizes (calls follow each other without a gap in a bench run)
zs = [rows[i]["Grid_Size_Z"] for i in fronts]
period = next(p for p in range(1, len(zs) + 1) if all(zs[k] == zs[k % p] for k in range(len(zs))))
calls = fronts[::period]
i0 = calls[-back]
i1 = calls[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
busy = {}
for r in rows[i0:i1]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r["Stream_Id"]
    busy[q] = busy.get(q, 0) + e - s
    print(f"s{q:>3} {s:9.1f} {e - s:8.1f}  {name:28s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']} lds {r['LDS_Block_Size']} vgpr {r['VGPR_Count']}")
print("busy µs per stream:", {k: round(v) for k, v in busy.items()}, "span", round((max(int(r["End_Timestamp"]) for r in rows[i0:i1]) - t0) / 1e3))
