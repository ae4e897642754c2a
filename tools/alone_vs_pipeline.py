"""per-kernel HIP-event time per launch: in the pipeline vs with nothing overlapping (seq_serial) - 200 frames cfg2"""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
os.environ.setdefault("VSM_HOST_THREADS", "14")
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
fr = torch.from_numpy(host).cuda()
tr = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12]); trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher(); m.set_intrinsics(*[float(x) for x in g["intr"]])
for _ in range(3): m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
res = {}
for mode in (0, 1):
    m.set_option("seq_serial", mode)
    m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    m.set_profiling(True)
    for _ in range(3): m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    torch.cuda.synchronize()
    res[mode] = m.kernel_stats(); m.set_profiling(False)
print(f"{'kernel':28s} {'launches':>8s} {'pipeline us':>12s} {'alone us':>10s} {'ratio':>6s}   total per call: pipeline / alone (ms)")
tp = ta = 0
for k in res[0]:
    ms0, n0 = res[0][k]; ms1, n1 = res[1].get(k, (0, 0))
    if not n0: continue
    a, b = ms0 / n0 * 1e3, (ms1 / n1 * 1e3 if n1 else 0)
    tp += ms0 / 3; ta += ms1 / 3
    print(f"{k:28s} {n0 // 3:8d} {a:12.1f} {b:10.1f} {a / b if b else 0:6.2f}   {ms0 / 3:6.3f} / {ms1 / 3:6.3f}")
print("sum of kernel time per call: pipeline %.2f ms, alone %.2f ms" % (tp, ta))
