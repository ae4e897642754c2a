import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
P = vm.P_MATCH
rs = np.random.RandomState(5)

def make_list(n, w=1242, h=375, dup=0.02, method=2, grid=False):
    m = np.zeros(n, dtype=P)
    if grid:
        g = np.stack(np.meshgrid(np.arange(20, 20 + 2 * int(np.ceil(np.sqrt(n))), 2), np.arange(20, 20 + 2 * int(np.ceil(np.sqrt(n))), 2)), -1).reshape(-1, 2)[:n]
        u, v = g[:, 0], g[:, 1]
    else:
        u = rs.randint(6, w // 2 - 6, n) * 2
        v = rs.randint(6, h // 2 - 6, n) * 2
    nd = int(n * dup)
    if nd and n > 10:
        src = rs.randint(0, n, nd); dst = rs.randint(0, n, nd)
        u[dst] = u[src]; v[dst] = v[src]
    m["u1c"] = u; m["v1c"] = v
    fl = rs.randint(-3, 4, (n, 2)); bad = rs.rand(n) < 0.1
    fl[bad] += rs.randint(-30, 30, (bad.sum(), 2))
    m["u1p"] = u + 6 + fl[:, 0]; m["v1p"] = v + fl[:, 1]
    d = 20 + rs.randint(-2, 3, n); d[rs.rand(n) < 0.05] += 17
    m["u2c"] = u - d; m["v2c"] = v
    m["u2p"] = m["u1p"] - d - rs.randint(-1, 2, n); m["v2p"] = m["v1p"]
    for k in ("i1p", "i2p", "i1c", "i2c"):
        m[k] = rs.randint(0, 9000, n)
    return m

if __name__ == "__main__":
    ok = True
    for n in (0, 1, 3, 4, 5, 17, 100, 480, 481, 700, 961, 2000, 3000, 7400, 9000):
        for method in (0, 1, 2):
            for grid in (False, True):
                lst = make_list(n, grid=grid)
                hs, hr, _ = vm.remove_outliers(lst, method, 1242, 375)
                for gt in (False, True):
                    if gt and n > 8192: continue
                    gs, gr, us = vm.remove_outliers(lst, method, 1242, 375, gpu=True, gpu_ties=gt, copies=2)
                    same = len(hs) == len(gs) and hs.tobytes() == gs.tobytes() and np.array_equal(hr, gr)
                    if not same:
                        ok = False
                        print("MISMATCH n", n, "method", method, "grid", grid, "gpu_ties", gt, len(hs), len(gs), np.array_equal(hr, gr))
        print("n", n, "done", len(hs), flush=True)
    # all matches at one pixel, collinear
    for lst in (make_list(50, dup=0),):
        lst["u1c"] = 100; lst["v1c"] = 50
        hs, hr, _ = vm.remove_outliers(lst, 2, 1242, 375); gs, gr, _ = vm.remove_outliers(lst, 2, 1242, 375, gpu=True)
        print("all equal", len(hs), len(gs), hs.tobytes() == gs.tobytes())
        lst = make_list(300, dup=0); lst["v1c"] = 40
        hs, hr, _ = vm.remove_outliers(lst, 2, 1242, 375); gs, gr, _ = vm.remove_outliers(lst, 2, 1242, 375, gpu=True)
        print("collinear", len(hs), len(gs), hs.tobytes() == gs.tobytes())
    lst = make_list(7400)
    for copies in (1, 50):
        gs, gr, us = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=False, copies=copies)
        print("7400 x", copies, "chain us", us)
    print("ALL OK" if ok else "FAILURES")

