import os, sys, time, importlib
ROOT = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, ROOT)
import bench  # pin + env
import numpy as np, torch
def stat():
    d = {}
    for l in open("/sys/fs/cgroup/cpu.stat"):
        k, v = l.split(); d[k] = int(v)
    return d
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
g = np.load(os.path.join(ROOT, "tests/golden/cfg4_seq200_tr_8seeds.npz"))
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
fr = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])).cuda()
tr = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12]); trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher(); m.set_intrinsics(*[float(x) for x in g["intr"]])
def threads():
    d = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            d[tid] = (name, int(rest[11]), int(rest[12]), int(rest[9]))   # utime, stime (ticks of 10 ms), minflt... (field 10 = minflt)
        except Exception:
            pass
    return d
s0 = stat()
for i in range(40):
    ta = threads()
    a = stat(); t = time.perf_counter()
    m.run_sequence(fr[:, 0], fr[:, 1], 2, tr, trv, fetch=False)
    dt = (time.perf_counter() - t) * 1e3; b = stat()
    tb = threads()
    if dt > 6 or i < 2:
        for tid, (name, ut, st, mf) in sorted(tb.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - sum(ta.get(kv[0], ("", 0, 0, 0))[1:3])))[:8]:
            o = ta.get(tid, (name, 0, 0, 0))
            if ut + st - o[1] - o[2] > 0 or mf - o[3] > 100:
                print("    thread %s %-16s utime +%d stime +%d ticks, minor faults +%d" % (tid, name, ut - o[1], st - o[2], mf - o[3]))
        print("call %d: %.2f ms, throttled periods +%d, throttled_usec +%d, usage_usec +%d" % (i, dt, b["nr_throttled"] - a["nr_throttled"], b["throttled_usec"] - a["throttled_usec"], b["usage_usec"] - a["usage_usec"]))
e = stat()
print("whole loop: nr_periods +%d nr_throttled +%d throttled_usec +%d" % (e["nr_periods"] - s0["nr_periods"], e["nr_throttled"] - s0["nr_throttled"], e["throttled_usec"] - s0["throttled_usec"]))
