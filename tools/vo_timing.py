"""live stereo VO (vsm_vo_stereo_process_device per frame): where a frame's time goes (mean us over frames 5..)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, int(os.environ.get("VSM_TOOL_FRAMES", 120))
cv = synth.canvas(1234, W, H)
frames = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
intr = [float(x) for x in g["intr"]]
other = None
if os.environ.get("VSM_TOOL_OTHER_HANDLE"):  # a second handle alive beside the VO's, as in bench.py (its look-ahead context, pool and poller)
    nl = 200 if os.environ["VSM_TOOL_OTHER_HANDLE"] == "lookahead" else 0
    other = vm.Matcher()
    other.set_intrinsics(*intr)
    if nl:
        fr = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nl)])).cuda()
        g2 = g
        tr12 = np.ascontiguousarray(g2["s1234_tr_in"][:nl].reshape(nl, 16)[:, :12])
        trv = np.ascontiguousarray(g2["s1234_tr_valid"][:nl].astype(np.uint8))
        for _ in range(3):
            other.run_sequence(fr[:, 0], fr[:, 1], 2, tr12, trv, fetch=False)
    else:
        for f in range(10):
            other.push_back(frames[f, 0], frames[f, 1])
            other.match_features(2, None)
    torch.cuda.synchronize()
if os.environ.get("VSM_TOOL_NOGC"):  # (a full collection of the interpreter's objects - torch and numpy are loaded - is a 60 ms pause)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
for rep in range(2):
    vm.vo_sampler_seed(71)
    vo = vm.VisualOdometryStereo(*intr)
    if os.environ.get("VSM_TOOL_PIN_CALLER") and vm.forkjoin_cpus():
        os.sched_setaffinity(0, vm.forkjoin_cpus())
    T, calls = [], []
    for f in range(nf):
        t0 = time.perf_counter()
        vo.process(frames[f, 0], frames[f, 1])
        calls.append((time.perf_counter() - t0) * 1e6)
        T.append(list(vo.timings().values()) if isinstance(vo.timings(), dict) else list(vo.timings()))
    print("rep", rep, "process call %.0f us (p50 %.0f, p90 %.0f, max %.0f);" % (np.mean(calls[8:]), np.percentile(calls[8:], 50), np.percentile(calls[8:], 90), max(calls[8:])), vo.timings().keys() if isinstance(vo.timings(), dict) else "", np.round(np.array(T)[5:].mean(0), 0))
    print("   slowest calls (frame, us):", [(int(i), int(calls[i])) for i in np.argsort(calls)[::-1][:5]], "the slowest one's split:", np.round(T[int(np.argmax(calls))], 0))
    vo.close()
