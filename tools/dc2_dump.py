import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
from dc2_check import make_list
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
verbose = len(sys.argv) > 2
lst = make_list(n, dup=0)
os.environ["VSM_DC2_DUMP"] = "/tmp/dc2.bin"
try:
    gs, gr, us = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=False, copies=1)
except Exception as e:
    print("ERR", e)
raw = np.fromfile("/tmp/dc2.bin", dtype=np.int32)
m, nn = raw[0], raw[1]
tri = raw[2:2 + m * 16].reshape(2 * m, 8)
pt = raw[2 + m * 16:2 + m * 17].view(np.uint32)
idv = raw[2 + m * 17:2 + m * 18]
print("m", m, "n", nn)
pts = np.stack([lst["u1c"], lst["v1c"]], 1).astype(np.int64)
# consistency of pt / id
px, py = pt & 0xffff, pt >> 16
okpt = np.array_equal(px, pts[idv, 0]) and np.array_equal(py, pts[idv, 1])
print("pt/id consistent:", okpt, "ids distinct:", len(set(idv.tolist())) == m)
v = tri[:, 4:7]
real = (v >= 0).all(1)
print("real triangles", real.sum(), "ghost", ((v < 0).any(1) & ~(v < 0).all(1)).sum(), "unused", (v < 0).all(1).sum())
def canon(t):
    t = np.sort(np.asarray(t), axis=1)
    return t[np.lexsort(t.T[::-1])]
np.set_printoptions(linewidth=200)
if verbose:
    print(tri[:60])
    print("pts by position", list(zip(px.tolist(), py.tolist())), "ids", idv.tolist())
    print("host", vm.host_delaunay(pts, 1).tolist())
print("vertex max per word", tri[:, 4].max(), tri[:, 5].max(), tri[:, 6].max(), "words 3/7 values", set(tri[:, 3].tolist()), set(tri[:, 7].tolist()))
real &= (v < m).all(1)
dev = canon(idv[v[real]])
host = canon(vm.host_delaunay(pts, 1))
print("host triangles", len(host), "equal:", dev.shape == host.shape and np.array_equal(dev, host))
if dev.shape != host.shape or not np.array_equal(dev, host):
    hs = set(map(tuple, host.tolist())); ds = set(map(tuple, dev.tolist()))
    print("only host", len(hs - ds), "only dev", len(ds - hs))
    print("range of vertex idx", v[real].min() if real.any() else None, v[real].max() if real.any() else None)
    bad = (v >= m).any(1)
    print("records with vertex >= m:", bad.sum(), "neighbour range", tri[:, :3].min(), tri[:, :3].max())
    print(tri[:12])
