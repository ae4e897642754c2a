"""Every launch's HIP-event span of a few main-stream kernels inside the otherwise unprofiled look-ahead pipeline (spans around
ONE kernel at a time: the spans' own event records are packets too).  python tools/span_probe.py [kernel names]"""
import importlib, os, sys
ROOT="/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
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
m = vm.Matcher(); m.set_intrinsics(*[float(x) for x in g["intr"]])
for i in range(3): m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
for only in (sys.argv[1:] or ["k_match<16>:pass2", "k_refine", "k_compact_matches:pass2"]):
    print(only, flush=True)
    m.set_profiling(True, only=only, print_spans=True)
    for i in range(3): m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    torch.cuda.synchronize(); m.set_profiling(False)
