"""What the Delaunay chains cost k_match: life of every wave of the dense pass, launch by launch, in the pipeline and with
the GPU to itself (seq_serial).  Library built with tools/build_variant.sh NAME -DVSM_MATCH_TIMING=3, VSM_LIB_PATH set.

A launch that takes longer beside the chains either has waves that LIVE longer (they share issue slots / memory pipes with
the chains' waves) or waves that live as long as ever but fewer of them at once (the chains hold the registers / LDS / wave
slots the dispatcher needs to place them).  Per launch: span (first wave start .. last wave end), mean and p90 wave life,
and resident waves per SIMD = sum of lives / span / (256 CUs x 4 SIMDs)."""
import ctypes
import importlib
import os
import sys

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
tr12 = np.ascontiguousarray(g["s1234_tr_in"][:nf].reshape(nf, 16)[:, :12])
trv = np.ascontiguousarray(g["s1234_tr_valid"][:nf].astype(np.uint8))
Lb = ctypes.CDLL(os.environ["VSM_LIB_PATH"])
buf = (ctypes.c_uint * (8 << 18))()
TICK_US = 0.01  # s_memrealtime: 100 MHz, device-wide


def run(serial):
    m = vm.Matcher()
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    if serial:
        m.set_option("seq_serial", 1)
    for _ in range(2):
        m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
    torch.cuda.synchronize()
    Lb.vsm_debug_match_timing(buf, 1 << 18, 1)
    m.run_sequence(frames[:, 0], frames[:, 1], 2, tr12, trv, fetch=False)
    torch.cuda.synchronize()
    n = Lb.vsm_debug_match_timing(buf, 1 << 18, 1)
    a = np.frombuffer(buf, dtype=np.uint32)[: 8 * n].reshape(n, 8).astype(np.int64).copy()
    start = a[:, 7]
    start = (start - start.min()) & 0xFFFFFFFF
    order = np.argsort(start)
    a, start = a[order], start[order]
    a[:, 0] = a[:, 6]  # life on the wall clock
    end = start + a[:, 0]
    # launches: a new one begins where a wave starts after every earlier wave has ended (the stream is in order)
    cuts = [0]
    run_end = end[0]
    for i in range(1, n):
        if start[i] > run_end:
            cuts.append(i)
        run_end = max(run_end, end[i])
    cuts.append(n)
    print("%s: %d waves in %d launches" % ("serial" if serial else "pipeline", n, len(cuts) - 1))
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        life = a[c0:c1, 0]
        span = end[c0:c1].max() - start[c0]
        # how many waves were alive, sampled over the span
        ts = np.linspace(start[c0], start[c0] + span, 41)[1:-1]
        alive = [(int(((start[c0:c1] <= t) & (end[c0:c1] > t)).sum())) for t in ts]
        print("  launch at %8.1f us: %6d waves, span %7.1f us, wave life mean %6.1f p50 %6.1f p90 %6.1f max %6.1f us, stages %s k cycles, resident waves/SIMD mean %.2f (min %.2f max %.2f over the span)"
              % (start[c0] * TICK_US, c1 - c0, span * TICK_US, life.mean() * TICK_US, np.percentile(life, 50) * TICK_US, np.percentile(life, 90) * TICK_US,
                 life.max() * TICK_US, "/".join("%.0f" % (a[c0:c1, k].mean() / 1e3) for k in (1, 2, 3, 4)), life.sum() / span / 1024.0, min(alive) / 1024.0, max(alive) / 1024.0))
    del m


run(True)
run(False)
