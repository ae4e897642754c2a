"""Triangle's vertex sort emulated on the host (vsm_host_ties): microseconds per list on one core."""
import ctypes as C, numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.environ.get("VSM_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'opencl-structure-from-motion_amd', 'libvisomatch.so')))
vp, i32 = C.c_void_p, C.c_int32
L.vsm_host_ties.argtypes = [vp, vp, i32, vp, i32]; L.vsm_host_ties.restype = i32
# (u1c, v1c) of a real pass-2 list of the benchmark sequence in list order (frame pair 1-2, from the oracle): the order matters -
# a list comes bin by bin, i.e. partly sorted, and Hoare's partition swaps little on it (random keys cost 30 % more)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
k = np.load(os.path.join(ROOT, "tools", "real_list_keys.npz"))
x, y = np.ascontiguousarray(k["x"]), np.ascontiguousarray(k["y"])
n = len(x)
if len(sys.argv) > 1 and sys.argv[1] == "random":
    rs = np.random.RandomState(5)
    p = rs.permutation(n); x, y = np.ascontiguousarray(x[p]), np.ascontiguousarray(y[p])
out = np.zeros(4096, dtype=np.int32)
def run(reps):
    t = time.perf_counter()
    for _ in range(reps):
        k = L.vsm_host_ties(x.ctypes.data, y.ctypes.data, n, out.ctypes.data, 2048)
    return (time.perf_counter() - t) / reps * 1e6, k
run(50)
best = min(run(200)[0] for _ in range(5))
print("vsm_host_ties: %.1f us per %d-key list, %d patches, checksum %d" % (best, n, run(1)[1], int(out[:2 * run(1)[1]].astype(np.int64).sum())))
