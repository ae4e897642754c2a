import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
from dc2_check import make_list
for n in (100, 200, 300, 480):
    for rep in range(3):
        lst = make_list(n)
        hs, hr, _ = vm.remove_outliers(lst, 2, 1242, 375)
        try:
            gs, gr, us = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=False, copies=2)
            print(n, rep, "same" if hs.tobytes() == gs.tobytes() else f"DIFF {len(hs)} {len(gs)}", flush=True)
        except Exception as e:
            print(n, rep, "ERR", e, flush=True)
