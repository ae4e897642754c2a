"""refinement = 2 through the look-ahead call (GPU-resident form) against the per-frame API: 100 frames 1242x375, quad matching"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 100
cv = synth.canvas(1234, W, H)
fr = [synth.stereo_frame(cv, f, W, H) for f in range(nf)]
L = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
R = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
m = vm.Matcher(refinement=2)
m.run_sequence(L, R, 2, fetch=False)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    got = m.run_sequence(L, R, 2, fetch=False)
dt = (time.perf_counter() - t) / 3
print("look-ahead form", m.sequence_path(), f"{nf / dt:9.1f} frame-pairs/s")
got = m.run_sequence(L, R, 2)
p = vm.Matcher(refinement=2)
t = time.perf_counter()
same = True
for f in range(nf):
    p.push_back(L[f], R[f])
    p.match_features(2)
    a = p.get_matches()
    same = same and len(a) == len(got[f]) and a.tobytes() == got[f].tobytes()
dt2 = time.perf_counter() - t
print("per frame", f"{nf / dt2:9.1f} frame-pairs/s", "same lists:", same, "matches in the last frame:", len(got[-1]))
