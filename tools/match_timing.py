"""life of the waves of k_match's dense pass (library built with tools/build_variant.sh NAME -DVSM_MATCH_TIMING, VSM_LIB_PATH set)"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 50
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
frames = torch.from_numpy(host).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
tr12 = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher()
m.set_intrinsics(*[float(x) for x in g["intr"]])
Lb = ctypes.CDLL(os.environ["VSM_LIB_PATH"])
buf = (ctypes.c_uint * (8 << 18))()
acc = (ctypes.c_ulonglong * 16)()
os.environ["VSM_SEQ_CHUNK"] = "50"
m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
Lb.vsm_debug_match_timing(buf, 1 << 18, 1)
Lb.vsm_debug_match_acc(acc, 1)
m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
torch.cuda.synchronize()
n = Lb.vsm_debug_match_timing(buf, 1 << 18, 1)
Lb.vsm_debug_match_acc(acc, 1)
A = [float(x) for x in acc]
a = np.frombuffer(buf, dtype=np.uint32)[: 8 * n].reshape(n, 8).astype(np.float64)
print("waves", n)
if A[3]:
    print("wave-level trips per wave (both passes' waves over the dense pass's count): findMatch %.1f, u-bin iterations %.1f, scan iterations %.1f, judge rounds %.1f"
          % (A[3] / n, A[0] / n, A[1] / n, A[2] / n))
if A[8]:
    print("dense pass: what a per-bin head record with N inline candidates could spare a wave (every lane's longest run of candidates <= N):")
    print("  stages whose groups share a bin's run (flow + prediction): %.0f wave-stages, N = 7: %.3f, N = 15: %.3f" % (A[8], A[9] / A[8], A[10] / A[8]))
    if A[11]:
        print("  stages whose lanes take a bin each (stereo): %.0f wave-stages, N = 3: %.3f, N = 7: %.3f, N = 15: %.3f" % (A[11], A[12] / A[11], A[13] / A[11], A[14] / A[11]))
names = ["life", "stage 1", "stage 2", "stage 3", "stage 4", "bins + scan", "judging"]
for k, nm in enumerate(names):
    c = a[:, k]
    print("  %-12s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f ticks" % (nm, c.mean(), np.percentile(c, 50), np.percentile(c, 90), c.max()))
