#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md / profiles/README.md (never bench.py's `value`):
  * per-frame API fed from pageable HOST buffers (PCIe-inclusive rate) at 1242x375,
  * config 5: 2048x1024 stereo, per-frame and look-ahead,
  * config 3: 640x480 mono, flow matching (per-frame API).
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "opencl-structure-from-motion_amd"
import torch  # noqa: E402

vm = importlib.import_module(PKG + ".visomatch")
synth = importlib.import_module(PKG + ".synth")


def per_frame(m, frames, method, reps=1):
    n = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        for l, r in frames:
            m.push_back(l, r)
            m.match_features(method)
            n += 1
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


out = {}
# 1. host buffers, 1242x375 quad
seq = synth.stereo_sequence(1234, 1242, 375, 60)
m = vm.Matcher()
per_frame(m, seq[:10], 2)
out["cfg2_host_buffers_per_frame_fps"] = round(per_frame(m, seq, 2), 2)
dev = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in seq]
out["cfg2_device_buffers_per_frame_fps"] = round(per_frame(m, dev, 2), 2)
out["cfg2_matches_last_frame"] = int(len(m.get_matches()))
m.close()
# 2. 2048x1024
seq5 = synth.stereo_sequence(1234, 2048, 1024, 12)
m = vm.Matcher()
dev5 = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in seq5]
per_frame(m, dev5[:3], 2)
out["cfg5_2048x1024_per_frame_fps"] = round(per_frame(m, dev5, 2), 2)
out["cfg5_matches_last_frame"] = int(len(m.get_matches()))
out["cfg5_timings_us"] = m.timings()
L = torch.from_numpy(np.stack([l for l, _ in seq5])).cuda()
R = torch.from_numpy(np.stack([r for _, r in seq5])).cuda()
m.run_sequence(L, R, 2, fetch=False)
t0 = time.perf_counter()
m.run_sequence(L, R, 2, fetch=False)
out["cfg5_2048x1024_lookahead_fps"] = round(len(seq5) / (time.perf_counter() - t0), 2)
m.close()
# 3. mono flow 640x480
seq3 = [(torch.from_numpy(x).cuda(), None) for x in synth.mono_sequence(1234, 640, 480, 60)]
m = vm.Matcher()
per_frame(m, seq3[:10], 0)
out["cfg3_640x480_mono_flow_per_frame_fps"] = round(per_frame(m, seq3, 0), 2)
out["cfg3_matches_last_frame"] = int(len(m.get_matches()))
m.close()
print(json.dumps(out))
