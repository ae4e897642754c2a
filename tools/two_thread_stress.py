"""two matchers in two threads, look-ahead runs at the same time, REPS times: every list against the reference's hashes
(prints the first mismatches: thread, repetition, frame, count got / wanted)"""
import importlib, os, sys, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
vm = importlib.import_module("opencl-structure-from-motion_amd.visomatch")
synth = importlib.import_module("opencl-structure-from-motion_amd.synth")
import golden_util as G
os.environ.setdefault("VSM_SEQ_CHUNK", "12")
g = G.load("cfg2_seq200_tr")
w, h, nf = int(g["w"]), int(g["h"]), 36
cv = synth.canvas(int(g["seed"]), w, h)
fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = []
def work(k):
    m = vm.Matcher()
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    for rep in range(REPS):
        out = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
        for f in range(nf):
            if len(out[f]) != int(g["counts"][f]) or G.sha(out[f]) != str(g["hashes"][f]):
                bad.append((k, rep, f, len(out[f]), int(g["counts"][f]), m.sequence_path()))
    m.close()
ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
for t in ts: t.start()
for t in ts: t.join()
print("mismatches:", len(bad), bad[:12])
