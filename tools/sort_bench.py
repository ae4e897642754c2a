import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
rs = np.random.RandomState(1)
pts = np.stack([rs.randint(3, 618, 7400) * 2, rs.randint(3, 184, 7400) * 2], 1)
for _ in range(300):
    vm.host_delaunay(pts, 1)
