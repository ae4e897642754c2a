"""does a re-created context leave the process slower?  resident calls (chunks of A), resident calls with chunks of B (the
context is re-created), resident calls with A again: python tools/recreate_probe.py [A B]"""
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
A, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (110, 80)


def timed(label, chunk, host=False, n=6):
    m.set_option("seq_chunk", chunk)
    a = (hl, hr) if host else (dl, dr)
    for _ in range(2):
        m.run_sequence(a[0], a[1], 2, tr12, trv, fetch=False)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        m.run_sequence(a[0], a[1], 2, tr12, trv, fetch=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %s" % (label, " ".join("%.2f" % t for t in ts)), flush=True)


timed("resident, chunks of %d" % A, A)
timed("resident, chunks of %d (re-created)" % B, B)
timed("resident, chunks of %d (re-created)" % A, A)
timed("host-fed, chunks of %d" % B, B, host=True)
timed("resident, chunks of %d" % A, A)
timed("host-fed, chunks of %d" % B, B, host=True)
