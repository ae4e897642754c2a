"""Do kernel launches on OTHER streams slow down the matcher's L2-reusing kernels?  The look-ahead call runs with
seq_serial = 1 (every kernel of the library with the GPU to itself, HIP-event times) while a background thread keeps
launching one-element kernels on a torch stream of its own - they occupy no compute unit to speak of, but every kernel
start is an acquire (cache invalidate) and every end a release.  Prints the per-kernel times with and without them."""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
import numpy as np, torch
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
g = np.load(os.path.join(ROOT, "tests/golden/cfg4_seq200_tr_8seeds.npz"))
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
fr = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])).cuda()
tr = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12]); trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher(options={"seq_serial": 1}); m.set_intrinsics(*[float(x) for x in g["intr"]])
stop = False
count = [0]
def spam(period_us):
    s = torch.cuda.Stream()
    x = torch.zeros(1, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            x.add_(1.0)
            count[0] += 1
            if period_us:
                t = time.perf_counter()
                while (time.perf_counter() - t) * 1e6 < period_us:
                    pass
def measure(label):
    m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    m.set_profiling(True)
    c0, t0 = count[0], time.perf_counter()
    for _ in range(2):
        m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = m.kernel_stats()
    m.set_profiling(False)
    keys = ["k_front", "k_filters<false>", "k_nms:sparse", "k_emit", "k_match<16>:pass1", "k_match<16>:pass2", "k_compact_matches:pass2", "k_refine"]
    print("%-44s" % label, " ".join("%s=%.0f" % (k.replace("k_", "").replace("<16>:pass", "").replace("_matches:pass", ""), st[k][0] / max(st[k][1], 1) * 1e3) for k in keys if k in st),
          "| side launches per ms: %.0f" % ((count[0] - c0) / (dt * 1e3)))
measure("alone")
for period in (0, 5, 10, 20, 100):
    stop = False
    th = threading.Thread(target=spam, args=(period,)); th.start()
    time.sleep(0.2)
    measure("beside one-element kernels, period %d us" % period)
    stop = True; th.join()
# ... beside event records only (markers: no kernel at all)
def spam_events(period_us):
    st = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(64)]
    i = 0
    while not stop:
        evs[i & 63].record(st)
        i += 1
        count[0] += 1
        if period_us:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < period_us:
                pass
for period in (0, 10):
    stop = False
    th = threading.Thread(target=spam_events, args=(period,)); th.start()
    time.sleep(0.2)
    measure("beside event records only, period %d us" % period)
    stop = True; th.join()
# ... and beside ONE long-running single-wave kernel at a time (Triangle's vertex sort of a 7.4 k list on one wave: ~3 ms per
# launch, a few launches per call): is it the launches, or the mere presence of another kernel in flight?
def long_waves():
    rs = np.random.RandomState(3)
    pts = np.stack([rs.randint(0, 620, 7400) * 2, rs.randint(0, 187, 7400) * 2], 1)
    while not stop:
        vm.ties(pts, gpu=True)
        count[0] += 1
stop = False
th = threading.Thread(target=long_waves); th.start()
time.sleep(0.2)
measure("beside one long single-wave kernel")
stop = True; th.join()
measure("alone again")
