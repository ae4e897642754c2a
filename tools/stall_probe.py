"""The once-per-process stall of the look-ahead call: what the operating system and the GPU driver counted around it.

Runs the call N times; before / after every call snapshots: wall time, process page faults and context switches
(getrusage, /proc/self/status), /proc/vmstat (THP collapse, TLB shootdowns, NUMA hints, migrations, compaction),
AnonHugePages of the process, and - where the driver shows them to an ordinary user - KFD's per-process eviction
time.  Prints the slowest call beside the median call.
  python tools/stall_probe.py [calls] [sleep_ms_between_calls]
"""
import glob
import importlib
import os
import resource
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 200
cv = synth.canvas(1234, W, H)
host = np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])
frames = torch.from_numpy(host).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
tr12 = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
m = vm.Matcher()
m.set_intrinsics(*[float(x) for x in g["intr"]])
L, R = frames[:, 0], frames[:, 1]

VM_KEYS = ["thp_collapse_alloc", "thp_fault_alloc", "thp_split_page", "nr_tlb_remote_flush", "nr_tlb_remote_flush_received",
           "nr_tlb_local_flush_all", "numa_hint_faults", "numa_pages_migrated", "pgmigrate_success", "compact_stall", "pgfault",
           "numa_pte_updates"]


def vmstat():
    d = {}
    try:
        for ln in open("/proc/vmstat"):
            k, v = ln.split()
            if k in VM_KEYS:
                d[k] = int(v)
    except OSError:
        pass
    return d


def status():
    d = {}
    for ln in open("/proc/self/status"):
        if ln.startswith(("voluntary_ctxt", "nonvoluntary_ctxt", "VmRSS", "VmData", "RssAnon")):
            k, v = ln.split(":")
            d[k] = int(v.split()[0])
    return d


def thread_stats():
    """sum over threads of utime + stime ticks, minor faults, and nonvoluntary switches"""
    tot = [0, 0, 0]
    for st in glob.glob("/proc/self/task/*/stat"):
        try:
            f = open(st).read().rsplit(")", 1)[1].split()
            tot[0] += int(f[11]) + int(f[12])   # utime + stime
            tot[1] += int(f[7])                 # minflt
        except (OSError, IndexError):
            pass
    for st in glob.glob("/proc/self/task/*/status"):
        try:
            for ln in open(st):
                if ln.startswith("nonvoluntary_ctxt"):
                    tot[2] += int(ln.split()[1])
        except OSError:
            pass
    return tot


def kfd():
    d = {}
    for p in glob.glob("/sys/class/kfd/kfd/proc/%d/stats_*/evicted_ms" % os.getpid()):
        try:
            d["evicted_ms"] = d.get("evicted_ms", 0) + int(open(p).read())
        except (OSError, ValueError):
            pass
    return d


def hugepages():
    try:
        for ln in open("/proc/self/smaps_rollup"):
            if ln.startswith("AnonHugePages"):
                return int(ln.split()[1])
    except OSError:
        pass
    return -1


def snap():
    r = resource.getrusage(resource.RUSAGE_SELF)
    s = {"minflt": r.ru_minflt, "majflt": r.ru_majflt, "nvcsw": r.ru_nvcsw, "nivcsw": r.ru_nivcsw, "utime_ms": r.ru_utime * 1e3,
         "stime_ms": r.ru_stime * 1e3, "anon_huge_kb": hugepages()}
    s.update(vmstat())
    s.update(status())
    s.update(kfd())
    t = thread_stats()
    s["thr_ticks"], s["thr_minflt"], s["thr_nonvol"] = t
    return s


def maps():
    d = {}
    for ln in open("/proc/self/maps"):
        f = ln.split()
        a, b = (int(x, 16) for x in f[0].split("-"))
        d[a] = (b - a, f[1], " ".join(f[5:]) if len(f) > 5 else "[anon]")
    return d


n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
pause = float(sys.argv[2]) / 1e3 if len(sys.argv) > 2 else 0.0
rows = []
t_first = None
for i in range(n):
    a = snap()
    ma = maps()
    t0 = time.perf_counter()
    if t_first is None:
        t_first = t0
    m.run_sequence(L, R, 2, tr12, trv, fetch=False)
    dt = (time.perf_counter() - t0) * 1e3
    b = snap()
    rows.append((i, (t0 - t_first) * 1e3, dt, {k: b[k] - a[k] for k in b if k in a}))
    if i < 3 or i == n - 1:
        mq = [v[0] for v in maps().values() if 150e6 < v[0] < 260e6 and v[2] == "[anon]"]
        print("  after call %d: %d anonymous mappings of 150-260 MB (one context-save area per hardware compute queue): %s; threads %d"
              % (i, len(mq), [round(x / 1e6, 1) for x in mq], len(os.listdir("/proc/self/task"))))
    if i >= 3 and b.get("VmData", 0) - a.get("VmData", 0) > 20000:
        print("  threads now %d" % len(os.listdir("/proc/self/task")))
        mb = maps()
        for k2, v in sorted(mb.items()):
            if k2 not in ma or ma[k2][0] != v[0]:
                print("  call %d (%.2f ms): mapping %x %+.1f MB now %.1f MB %s %s" % (i, dt, k2, (v[0] - ma.get(k2, (0,))[0]) / 1e6, v[0] / 1e6, v[1], v[2]))
        for k2, v in sorted(ma.items()):
            if k2 not in mb:
                print("  call %d: mapping %x gone (%.1f MB %s)" % (i, k2, v[0] / 1e6, v[2]))
    if pause:
        time.sleep(pause)
steady = rows[3:]
med = sorted(r[2] for r in steady)[len(steady) // 2]
worst = max(steady, key=lambda r: r[2])
typ = min(steady, key=lambda r: abs(r[2] - med))
print("calls %d, median %.2f ms, slowest %.2f ms (call %d, %.0f ms after the first call began); calls over 1.5 x median: %s"
      % (n, med, worst[2], worst[0], worst[1], [(r[0], round(r[1]), round(r[2], 1)) for r in steady if r[2] > 1.5 * med]))
keys = sorted(set(worst[3]) | set(typ[3]))
print("%-30s %14s %14s" % ("counter (delta over the call)", "slowest call", "median call"))
for k in keys:
    w, t = worst[3].get(k, 0), typ[3].get(k, 0)
    if w or t:
        print("%-30s %14.1f %14.1f" % (k, w, t))
print("kfd eviction counters visible:", bool(kfd()), "| transparent_hugepage:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
      if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "?",
      "| numa_balancing:", open("/proc/sys/kernel/numa_balancing").read().strip() if os.path.exists("/proc/sys/kernel/numa_balancing") else "?")
