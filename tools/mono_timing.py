"""wall-clock split of the monocular egomotion (vsm_vo_mono_process_matches) on a synthetic scene"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402
from oracle import bindings as B  # noqa: E402
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
rs = np.random.RandomState(7)
f, cu, cv = MG.KITTI["f"], MG.KITTI["cu"], MG.KITTI["cv"]
for n, it in ((1000, 2000), (5000, 2000)):
    m = MG.mono_scene(rs, n, (0.002, 0.012, -0.001, 0.03, -0.01, -0.9))
    kw = dict(height=1.65, pitch=-0.08, ransac_iters=it)
    v = vm.VisualOdometryMono(f, cu, cv, **kw)
    for rep in range(3):
        vm.vo_sampler_seed(71)
        t = time.perf_counter()
        ok, T = v.process_matches(m)
        dt = time.perf_counter() - t
    tm = v.timings()
    print("device svd", v.device_svd(), "n", n, "iters", it, "ok", ok, "inliers", v.get_number_of_inliers(), "GPU+host %.2f ms" % (dt * 1e3),
          "split us: F fits %.0f, inlier count %.0f, R|t+triangulation %.0f, plane vote %.0f" % tuple(tm[4:8]))
    v.close()
    if B.have_ref():
        r = B.RefMonoVO(f, cu, cv, **kw)
        t = time.perf_counter()
        ok_r, T_r = r.process_matches(m)
        dr = time.perf_counter() - t
        print("   reference (1 core): %.1f ms; Tr_delta equal: %s" % (dr * 1e3, "n/a (sampler state)"))
        r.close()
