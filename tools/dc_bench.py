"""GPU side of the exact Delaunay (kd order, sub-trees, merge levels): microseconds per 50 triangulations vs split"""
import importlib, os, sys
import numpy as np
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
rs = np.random.RandomState(1)
pts = np.stack([rs.randint(3, 618, 7400) * 2, rs.randint(3, 184, 7400) * 2], 1)
x = np.ascontiguousarray(pts[:, 0], dtype=np.int32); y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
for kd in (0, 1):
    for leaf, top in ((14, 0), (28, 0), (56, 0), (14, 120), (14, 240), (14, 480), (28, 240), (56, 240), (480, -1), (240, -1)):
        us = vm.lib().vsm_debug_dc_bench(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), leaf, top, kd, 50, 5)
        print("kd order on the GPU %d, sub-trees of <= %3d points, merge levels up to %5d points: %.0f us for 50 triangulations" % (kd, leaf, top, us))
