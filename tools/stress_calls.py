"""the look-ahead call a few hundred times in one process (chunk size from argv): every call's lists against the reference's
hashes; stops at the first failure"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5"); os.environ.setdefault("VSM_HOST_THREADS", "14")
import numpy as np, torch
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
g = np.load(os.path.join(ROOT, "tests/golden/cfg4_seq200_tr_8seeds.npz"))
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
fr = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])).cuda()
tr = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12]); trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = vm.Matcher(options={"seq_chunk": chunk} if chunk else None); m.set_intrinsics(*[float(x) for x in g["intr"]])
t0 = time.perf_counter()
worst = 0.0
for i in range(calls):
    t = time.perf_counter()
    m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    worst = max(worst, time.perf_counter() - t)
    if i % 25 == 0:
        ok = all(hashlib.sha256(m.sequence_matches(f).tobytes()).hexdigest() == str(g["s1234_hashes"][f]) for f in range(nf))
        print("call", i, "lists", "OK" if ok else "MISMATCH", "worst call so far %.1f ms" % (worst * 1e3), flush=True)
        if not ok:
            sys.exit(1)
print("done: %d calls, %.2f ms per call, worst %.1f ms" % (calls, (time.perf_counter() - t0) / calls * 1e3, worst * 1e3))
