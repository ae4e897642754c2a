#!/usr/bin/env python3
"""per-kernel averages of every counter found in rocprofv3 counter_collection.csv files under a directory"""
import collections
import csv
import glob
import sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
only = sys.argv[2:] 
for k in sorted(d):
    if k.startswith("__amd") or (only and not any(o in k for o in only)):
        continue
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d[k].items())), "n=%d" % len(next(iter(d[k].values()))))
