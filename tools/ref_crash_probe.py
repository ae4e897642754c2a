"""round 4's crash record (gpurun_out/r4/t1.log): the reference's removeOutliers (oracle/_ref/libvisoref.so) called inside a
process that holds the HIP runtime.  Repeats that: HIP first, then the reference on the chain test's lists.
  python tools/ref_crash_probe.py            (VISO_REF_SO + LD_PRELOAD=libasan.so: the sanitizer build)"""
import faulthandler
import importlib
import importlib.util
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
from oracle import bindings as B

spec = importlib.util.spec_from_file_location("dc2_check", os.path.join(ROOT, "tools", "dc2_check.py"))
dc2 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(dc2)
torch.zeros(4).cuda()
n_ok = 0
for rep in range(2):
    for n in (0, 3, 4, 5, 17, 100, 480, 481, 961, 2000, 4500, 7400, 9000):
        for method in (0, 1, 2):
            for grid in (False, True):
                lst = dc2.make_list(n, grid=grid)
                hs, hr, _ = vm.remove_outliers(lst, method, 1242, 375)                      # host code of the library
                gs, gr, _ = vm.remove_outliers(lst, method, 1242, 375, gpu=True, copies=2)  # device chain (HIP in use)
                sys.stderr.write("ref n=%d method=%d grid=%d ... " % (n, method, grid))
                sys.stderr.flush()
                want = B.remove_outliers("ref", lst, method)
                sys.stderr.write("ok %d\n" % len(want))
                assert want.tobytes() == hs.tobytes() == gs.tobytes()
                n_ok += 1
print("no crash:", n_ok, "lists")
