"""which threads burn CPU during look-ahead runs: per-thread CPU seconds (from /proc/self/task) over N runs"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
frames = torch.from_numpy(host).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
tr_in, tr_valid = g["s1234_tr_in"], g["s1234_tr_valid"]
m = vm.Matcher()
m.set_intrinsics(*[float(x) for x in g["intr"]])
tr12 = np.ascontiguousarray(tr_in[:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(tr_valid[:nf].astype(np.uint8))
left_d, right_d = frames[:, 0], frames[:, 1]
run = lambda: m.run_sequence(left_d, right_d, 2, tr12, trv, fetch=False)
for _ in range(3):
    run()

def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[t] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out

a = snap()
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for _ in range(N):
    run()
wall = time.perf_counter() - t0
b = snap()
rows = sorted(((b[t][1] - a.get(t, (0, 0))[1], b[t][0], t) for t in b), reverse=True)
print("wall %.3f s for %d runs (%.2f ms per run); CPU seconds per thread:" % (wall, N, wall / N * 1e3))
tot = 0
for cpu, comm, t in rows:
    tot += cpu
    if cpu > 0.005 * wall:
        print("  %-16s tid %s  %.3f s (%.0f %% of a CPU)" % (comm, t, cpu, 100 * cpu / wall))
print("  total %.2f s = %.1f CPUs" % (tot, tot / wall))
