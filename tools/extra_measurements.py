#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md / profiles/README.md (never bench.py's `value`):
  * per-frame API fed from pageable HOST buffers (PCIe-inclusive rate) at 1242x375,
  * the look-ahead API on a 1000-frame sequence (the bench's 200 frames five times over, replayed Tr_delta): the
    rate once the pipeline's fill and drain no longer count; two independent sequences at once on the one GPU,
  * config 5: 2048x1024 stereo, per-frame and look-ahead,
  * config 3: 640x480 mono, flow matching (per-frame API),
  * the street scene (synth.road_*, depth-dependent disparity/flow): whole VisualOdometryStereo::process
    and VisualOdometryMono::process per frame, with the recovered forward motion.
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (the rank's CPU share, queue cap and CPUs near its GPU, set the way the benchmark sets them - before HIP starts)
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
# 1b. long sequence through the look-ahead API
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
cvs = synth.canvas(1234, 1242, 375)
fr = [synth.stereo_frame(cvs, f, 1242, 375) for f in range(200)]
reps = 5
L = torch.from_numpy(np.stack([l for l, _ in fr] * reps)).cuda()
R = torch.from_numpy(np.stack([r for _, r in fr] * reps)).cuda()
tr = np.ascontiguousarray(np.tile(g["s1234_tr_in"][:200].reshape(200, 16)[:, :12], (reps, 1)))
trv = np.ascontiguousarray(np.tile(g["s1234_tr_valid"][:200].astype(np.uint8), reps))
m = vm.Matcher()
m.set_intrinsics(*[float(x) for x in g["intr"]])
m.run_sequence(L, R, 2, tr, trv, fetch=False)
t0 = time.perf_counter()
for _ in range(3):
    m.run_sequence(L, R, 2, tr, trv, fetch=False)
out["cfg2_lookahead_1000_frames_fps"] = round(3 * 200 * reps / (time.perf_counter() - t0), 1)
m.close()
del L, R
# 1c. two independent 200-frame sequences at once on the one GPU (two handles, two caller threads, half the host threads
#     each): one sequence's pipeline fills while the other's drains
import threading
nt_all = int(os.environ.get("VSM_HOST_THREADS", "16"))
os.environ["VSM_HOST_THREADS"] = str(max(1, nt_all // 2))
L2 = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
R2 = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
tr2 = np.ascontiguousarray(g["s1234_tr_in"][:200].reshape(200, 16)[:, :12])
trv2 = np.ascontiguousarray(g["s1234_tr_valid"][:200].astype(np.uint8))
ms = [vm.Matcher() for _ in range(2)]
for mm in ms:
    mm.set_intrinsics(*[float(x) for x in g["intr"]])
    mm.run_sequence(L2, R2, 2, tr2, trv2, fetch=False)
reps2 = 20
def _loop(mm):
    for _ in range(reps2):
        mm.run_sequence(L2, R2, 2, tr2, trv2, fetch=False)
ths = [threading.Thread(target=_loop, args=(mm,)) for mm in ms]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
out["cfg2_two_sequences_at_once_fps"] = round(2 * reps2 * 200 / (time.perf_counter() - t0), 1)
for mm in ms:
    mm.close()
os.environ["VSM_HOST_THREADS"] = str(nt_all)
del L2, R2
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
# 4. street scene: live stereo VO and mono VO loops on images with real depth structure
pyr = synth.road_pyramid(1234)
W4, H4, n4 = 1242, 375, 40
road = [synth.road_stereo_frame(pyr, f, W4, H4) for f in range(n4)]
cu, cv = W4 // 2, (H4 * 2) // 5
droad = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in road]
for rep in range(2):
    vm.vo_sampler_seed(71)
    vo = vm.VisualOdometryStereo(float(synth.ROAD_F), float(cu), float(cv), 0.54)
    t0 = time.perf_counter()
    res = [vo.process(l, r) for l, r in droad]
    dt = time.perf_counter() - t0
    out["road_stereo_vo_fps"] = round(n4 / dt, 2)
    out["road_stereo_vo_success"] = int(sum(r[0] for r in res))
    out["road_stereo_vo_tz_mean"] = round(float(np.mean([r[3][2, 3] for r in res[1:]])), 4)
    out["road_stereo_matches_last_frame"] = int(vm.lib().vsm_num_matches(vm.lib().vsm_vo_stereo_matcher(vo.h)))
    vo.close()
for rep in range(2):
    vm.vo_sampler_seed(71)
    mo = vm.VisualOdometryMono(float(synth.ROAD_F), float(cu), float(cv), height=1.65, pitch=0.0)
    t0 = time.perf_counter()
    res = [mo.process(l) for l, _ in droad]
    dt = time.perf_counter() - t0
    out["road_mono_vo_fps"] = round(n4 / dt, 2)
    out["road_mono_vo_success"] = int(sum(r[0] for r in res))
    out["road_mono_vo_tz_mean"] = round(float(np.mean([r[1][2, 3] for r in res[1:]])), 4)
    out["road_mono_timings_us"] = [round(float(x), 1) for x in mo.timings()[:8]]
    mo.close()
print(json.dumps(out))
