"""the last look-ahead call of a rocprofv3 kernel trace as a timeline: python tools/trace_last_call.py <kernel_trace.csv> [gap_ms]
(calls are separated by gaps without kernels; per kernel: start / end in us from the call's first kernel, queue, name)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.4e6
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:44], r.get("Grid_Size_X", "")) for r in rows))
calls, cur, last_end = [], [], None
for e in ev:
    if last_end is not None and e[0] - last_end > gap:
        calls.append(cur)
        cur = []
    cur.append(e)
    last_end = max(last_end or 0, e[1])
calls.append(cur)
c = calls[-1]
t0 = c[0][0]
print("calls", len(calls), "kernels in the last", len(c))
for s, e, q, n, g in c:
    print("%8.0f %8.0f %7.0f q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
