"""cycles per phase of k_dc2_block / k_dc2_merge (library built by tools/build_variant_dc.sh NAME -DDC2_PHASE_TIMING, VSM_LIB_PATH set)"""
import ctypes
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
L = ctypes.CDLL(os.environ["VSM_LIB_PATH"])
buf = (ctypes.c_ulonglong * 256)()
vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=(n <= 8192), copies=copies)
L.vsm_debug_dc2_phases(buf, 1)
vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=(n <= 8192), copies=copies)
L.vsm_debug_dc2_phases(buf, 1)
a = np.array(buf[:], dtype=np.float64).reshape(16, 16)
MHZ = 100.0  # clock64() ticks per microsecond (s_memtime: constant 100 MHz on gfx9)
print("k_dc2_block: blocks", int(a[0, 0]))
names = {1: "init", 2: "leaves", 13: "write-out", 14: "total"}
names.update({3 + L: f"L{L}" for L in range(10)})
for k in sorted(names):
    if a[0, k]:
        print(f"  {names[k]:10s} {a[0, k] / max(a[0, 0], 1) / MHZ:9.2f} ticks per block")
for lv in range(0, 9):
    if a[1 + lv, 0]:
        c = a[1 + lv, 0]
        print(f"k_dc2_merge level {lv}: nodes {int(c)}  load {a[1 + lv, 1] / c / MHZ:8.2f} us  zip {a[1 + lv, 2] / c / MHZ:8.2f} us  band lines {a[1 + lv, 3] / c:8.0f} (max {int(a[1 + lv, 6])}, points max {int(a[1 + lv, 7])})  walks that left the band {int(a[1 + lv, 4])}  nodes done in global memory {int(a[1 + lv, 5])}")
if a[12, 0]:
    c = a[12, 0]
    print(f"k_dc2_prepare: lists {int(c)}  (x,y) radix sort {a[12, 1] / c:9.0f}  duplicates {a[12, 2] / c:9.0f}  y order {a[12, 3] / c:9.0f}  kd levels {a[12, 4] / c:9.0f}  clock64 ticks per list")
