"""Work counters of one representative frame pair (frames 0, 1) of BASELINE.json configs[4] (2048x1024 stereo, the 20 k and the
40 k dense-feature variants) from the oracle: what bench.py prices the matching / refinement kernels' algorithmic bytes
with for those configurations (same counters as the headline's cpu_baseline leg takes live).  CPU only.
  python tests/golden/make_cfg5_work_counters.py  ->  tests/golden/cfg5_work_counters.json"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings as B  # noqa: E402

synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
out = {}
for name in ("cfg5_2048x1024_quad_20k", "cfg5_2048x1024_quad"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    w, h = int(g["w"]), int(g["h"])
    seq = synth.stereo_sequence(int(g["seed"]), w, h, 2, blur=int(g["blur"]))
    om = B.CpuMatcher("oracle")
    for l, r in seq:
        om.push_back(l, r)
        om.match(2)
    c = om.counters()
    out[name] = dict(Q1=int(c["Q1"]), C1=int(c["C1"]), S1=int(c["S1"]), Q2=int(c["Q"] - c["Q1"]), C2=int(c["C"] - c["C1"]),
                     S2=int(c["S"] - c["S1"]), M1=int(len(om.stage(0))), M=int(c["M"]),
                     N=int(len(om.features("1c1")) + len(om.features("1c2"))), w=w, h=h,
                     final_matches_frame1=int(g["counts"][1][-1]))
    om.close()
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "cfg5_work_counters.json"), "w"), indent=1)
print(json.dumps(out))
