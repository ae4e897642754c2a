"""GPU sub-tree kernel of the exact Delaunay: microseconds per launch of 50 triangulations vs sub-tree size"""
import importlib, os, sys
import numpy as np
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
rs = np.random.RandomState(1)
pts = np.stack([rs.randint(3, 618, 7400) * 2, rs.randint(3, 184, 7400) * 2], 1)
x = np.ascontiguousarray(pts[:, 0], dtype=np.int32); y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
for leaf in (3, 7, 14, 28, 56, 112, 225):
    us = vm.lib().vsm_debug_dc_bench(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), leaf, 50, 5)
    print("sub-trees of <= %3d points: %.0f us per launch of 50 triangulations" % (leaf, us))
