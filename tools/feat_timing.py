"""cycles per phase of the fused image-side kernels, per wave (library built with
tools/build_variant.sh NAME -DVSM_FEAT_TIMING, VSM_LIB_PATH set): python tools/feat_timing.py [serial]"""
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
W, H, nf = 1242, 375, 100
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
frames = torch.from_numpy(host).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
tr12 = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher(options={"seq_serial": 1} if "serial" in sys.argv else None)
m.set_intrinsics(*[float(x) for x in g["intr"]])
Lb = ctypes.CDLL(os.environ["VSM_LIB_PATH"])
rec = np.zeros((5, 1 << 16, 10), dtype=np.uint32)
m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
torch.cuda.synchronize()
Lb.vsm_debug_feat_rec(rec.ctypes.data_as(ctypes.c_void_p), 0)
for k, name, ph in ((4, "k_front", ["fill", "barrier", "copy + half image", "Sobel + stores"]),
                    (0, "k_feat_dense", ["fill", "barrier", "patches", "barrier", "suppression"]),
                    (1, "k_feat_sparse", ["fill", "barrier", "patches", "barrier", "suppression f1", "f2 -> LDS", "suppression f2"]),
                    (3, "k_feat_scan", ["zero + barrier", "cells + histogram", "scan", "offsets + bin sums", "scan", "bin starts"]),
                    (2, "k_feat_order", ["requests", "barriers", "lists 0", "barrier + records 0 + barriers", "lists 1", "barrier + records 1"])):
    r = rec[k][rec[k][:, 0] != 0].astype(np.float64)
    if not len(r):
        continue
    print("%s: %d waves; cycles per wave, mean (p50, p90):" % (name, len(r)))
    for q, p in enumerate(ph):
        c = r[:, 1 + q]
        print("   %-16s %8.0f (%8.0f %8.0f)" % (p, c.mean(), np.percentile(c, 50), np.percentile(c, 90)))
    print("   %-16s %8.0f" % ("life", r[:, 1:1 + len(ph)].sum(axis=1).mean()))
