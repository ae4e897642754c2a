"""does a hipFree elsewhere in the process slow the look-ahead call down?  resident calls, a torch tensor of N MB allocated and
given back (torch.cuda.empty_cache -> hipFree), resident calls again: python tools/hipfree_probe.py [MB ...]"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
fr = [synth.stereo_frame(cv, f, W, H) for f in range(nf)]
hl = np.ascontiguousarray(np.stack([l for l, _ in fr]))
hr = np.ascontiguousarray(np.stack([r for _, r in fr]))
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
tr12 = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
dl, dr = torch.from_numpy(hl).cuda(), torch.from_numpy(hr).cuda()
m = vm.Matcher()
m.set_intrinsics(*[float(x) for x in g["intr"]])


def timed(label, n=6):
    for _ in range(2):
        m.run_sequence(dl, dr, 2, tr12, trv, fetch=False)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        m.run_sequence(dl, dr, 2, tr12, trv, fetch=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-60s %s" % (label, " ".join("%.2f" % t for t in ts)), flush=True)


timed("resident")
for mb in [int(a) for a in sys.argv[1:]] or [1, 64, 512, 1024, 2048, 4096]:
    x = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    x.zero_()
    torch.cuda.synchronize()
    del x
    t_free = time.perf_counter()
    torch.cuda.empty_cache()
    t_freed = time.perf_counter()
    # calls until one is back to normal: how long the slow spell lasts
    ts, t_back = [], None
    for _ in range(120):
        t0 = time.perf_counter()
        m.run_sequence(dl, dr, 2, tr12, trv, fetch=False)
        ts.append((time.perf_counter() - t0) * 1e3)
        if len(ts) >= 2 and ts[-1] < 4.4 and ts[-2] < 4.4:
            t_back = (t0 - t_freed) * 1e3
            break
    print("hipFree of %5d MB took %.1f ms; calls after it: %s%s" % (mb, (t_freed - t_free) * 1e3, " ".join("%.1f" % t for t in ts[:8]),
          " ... back to normal after %.0f ms (%d calls)" % (t_back, len(ts)) if t_back is not None else " ... never within 120 calls"), flush=True)
m2 = vm.Matcher()
m2.set_intrinsics(*[float(x) for x in g["intr"]])
m2.push_back(dl[0], dr[0])
m2.push_back(dl[1], dr[1])
m2.match_features(2, None)
timed("resident beside a second handle (per-frame ring)")
m2.close()
timed("resident after that handle's close")
