"""live stereo VO (vsm_vo_stereo_process_device per frame): where a frame's time goes (mean us over frames 5..)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
W, H, nf = 1242, 375, 120
cv = synth.canvas(1234, W, H)
frames = torch.from_numpy(np.stack([np.stack(synth.stereo_frame(cv, f, W, H)) for f in range(nf)])).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg4_seq200_tr_8seeds.npz"))
intr = [float(x) for x in g["intr"]]
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
    print("rep", rep, "process call %.0f us;" % np.mean(calls[5:]), vo.timings().keys() if isinstance(vo.timings(), dict) else "", np.round(np.array(T)[5:].mean(0), 0))
    vo.close()
