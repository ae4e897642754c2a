"""kernel times of the GPU-resident removeOutliers chain on `copies` lists at once (run under rocprofv3 --kernel-trace --stats)"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dc2_check import make_list  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7400
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lst = make_list(n)
for r in range(4):
    gs, gr, us = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=True, copies=copies)
    print(n, copies, "chain us (device vertex sort included)", us, flush=True)
