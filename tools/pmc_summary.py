#!/usr/bin/env python3
"""Condenses rocprofv3 PMC passes into profiles/<tag>_pmc_hbm.csv.

  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv>

The two inputs come from SEPARATE runs (`rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and
`--pmc WRITE_SIZE ...`; FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-per-frame`.
traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch: on gfx950 FETCH_SIZE reports half
of the bytes of wide streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
`bench_name` is the kernel's name in bench.py's kernel tables.
"""
import collections
import csv
import sys

# (k_match<G, BYBIN>: BYBIN = the launches without prior boxes, i.e. the first pass)
NAMES = {"k_match<2, false>": "k_match<16>:pass2", "k_match<4, false>": "k_match<16>:pass2", "k_match<8, false>": "k_match<16>:pass2",
         "k_match<2, true>": "k_match<16>:pass1", "k_match<4, true>": "k_match<16>:pass1", "k_match<8, true>": "k_match<16>:pass1", "k_nms_tile": "k_nms:dense",
         "k_nms_tile8": "k_nms:sparse", "k_nms_fixed<3, 16, 8, 1>": "k_nms:dense", "k_nms_fixed<9, 4, 4, 8>": "k_nms:sparse",
         "k_compact_write": "k_compact_matches", "k_compact_quad": "k_compact_matches", "k_compact_quad<0>": "k_compact_matches", "k_compact_quad<1>": "k_compact_matches", "k_compact_quad<2>": "k_compact_matches", "k_refine<true>": "k_refine", "k_refine<false>": "k_refine",
         "k_dc2_block": "k_dc_block", "k_dc2_merge": "k_dc_merge", "k_dc2_prepare": "k_dc_prepare_kd_order", "k_dc2_keys": "k_dc_keys",
         "k_dc2_ties": "k_dc_vertex_sort", "k_dc2_support": "k_dc_support", "k_dc2_support_lds": "k_dc_support", "k_dc2_export": "k_export_list", "k_dc2_compact": "k_dc_compact", "k_dc2_prior": "k_dc_prior"}


def bench_name(k):
    import re
    m = re.match(r"k_match<\d+, (false|true)", k)  # (k_match<G, BYBIN, HEADS>)
    if m:
        return "k_match<16>:pass1" if m.group(1) == "true" else "k_match<16>:pass2"
    return NAMES.get(k, k)


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


f = agg(sys.argv[1], "FETCH_SIZE")
w = agg(sys.argv[2], "WRITE_SIZE")
with open(sys.argv[3], "w", newline="") as o:
    wr = csv.writer(o)  # kernel names contain commas (template arguments)
    wr.writerow(["kernel", "bench_name", "avg_FETCH_SIZE_raw_KB", "avg_WRITE_SIZE_raw_KB", "dispatches",
                 "traffic_bytes_per_launch"])
    for k in sorted(set(f) | set(w)):
        if k.startswith("__amd"):
            continue
        a, b = f.get(k, (0, 0)), w.get(k, (0, 0))
        wr.writerow([k, bench_name(k), f"{a[0]:.2f}", f"{b[0]:.2f}", a[1], f"{(2 * a[0] + b[0]) * 1024:.0f}"])
print("wrote", sys.argv[3])
