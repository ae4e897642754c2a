"""per-frame wall-clock split of vsm_vo_stereo_process on the bench sequence (mean over frames, us)"""
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
W, H, nf = 1242, 375, int(sys.argv[1]) if len(sys.argv) > 1 else 200
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
frames = torch.from_numpy(host).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_seq200_ego.npz"))
intr = [float(x) for x in g["intr"]]
for rep in range(3):
    vm.vo_sampler_seed(71)
    vo = vm.VisualOdometryStereo(*intr)
    T = []
    t0 = time.perf_counter()
    for f in range(nf):
        vo.process(frames[f, 0], frames[f, 1])
        T.append(vo.timings())
    dt = time.perf_counter() - t0
    T = np.array(T)[2:]
    print("rep", rep, "fps %.1f" % (nf / dt), "match %.0f bucket+copy %.0f ego %.0f total-after-push %.0f us" % tuple(T.mean(0)),
          "inliers", vo.get_number_of_inliers(), "of", vo.get_number_of_matches())
    vo.close()
