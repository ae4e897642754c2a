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
for leaf, top in ((3, 0), (7, 0), (14, 0), (28, 0), (56, 0), (112, 0), (225, 0),
                  (14, 30), (14, 60), (14, 120), (14, 240), (14, 480), (14, 960), (14, 1900), (14, 10000),
                  (28, 480), (56, 120), (56, 240), (56, 480), (56, 960), (7, 480), (3, 480)):
    us = vm.lib().vsm_debug_dc_bench(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), leaf, top, 50, 5)
    print("sub-trees of <= %3d points, merge levels up to %5d points: %.0f us for 50 triangulations" % (leaf, top, us))
