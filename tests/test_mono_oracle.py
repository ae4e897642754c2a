"""The oracle's monocular egomotion (oracle/viso_mono_oracle.c: the reference's Matrix::svd / det,
8-point fundamental matrices, Sampson inliers, E -> R|t, triangulation, ground-plane vote) against
golden vectors recorded from the real reference and, where oracle/_ref is built, against the reference
itself in a fresh process (shared RANSAC sampler, see test_ego_oracle.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util as G

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def test_mono_cases_golden(B):
    G.replay_mono_cases(load("mono_cases"), B.OracleMonoVO, B.oracle_sampler_seed)


def test_mono_sequence_golden(B, synth):
    G.replay_mono_sequence(load("mono_seq12_640x480"), synth, B.OracleMonoVO, B.oracle_sampler_seed)


def test_svd_properties(B):
    """the restated SVD is a valid decomposition with the reference's conventions (descending order)"""
    rs = np.random.RandomState(0)
    for m, n in [(8, 9), (3, 3), (4, 4), (40, 9)]:
        A = rs.normal(size=(m, n))
        U, W, V = B.oracle_svd(A)
        r = min(m, n)
        assert np.all(np.diff(W) <= 0) and np.all(W >= 0)
        assert np.allclose(U[:, :r] @ np.diag(W) @ V[:, :r].T, A, atol=1e-9)
        assert np.allclose(V.T @ V, np.eye(n), atol=1e-9)


_LIVE = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from oracle import bindings as B
sys.path.insert(0, {golden!r})
import make_golden as MG
rs = np.random.RandomState({seed})
bad = 0
for (m_, n_) in [(8, 9), (3, 3), (4, 4), (60, 9), (9, 8), (3, 4)]:
    for t in range(25):
        A = rs.normal(size=(m_, n_)) * 10 ** rs.uniform(-3, 3)
        if t % 5 == 0: A[:, -1] = A[:, 0]
        if t % 7 == 0: A[1] = 0
        if not all(x.tobytes() == y.tobytes() for x, y in zip(B.ref_svd(A), B.oracle_svd(A))): bad += 1; print("svd", m_, n_, t)
for t in range(60):
    A = rs.normal(size=(3, 3))
    if t % 6 == 0: A[2] = A[0]
    if B.ref_det(A) != B.oracle_det(A): bad += 1; print("det", t)
f, cu, cv = MG.KITTI["f"], MG.KITTI["cu"], MG.KITTI["cv"]
for case in range({cases}):
    n = int(rs.choice([10, 11, 40, 150, 500]))
    motion = (rs.uniform(-.01, .01), rs.uniform(-.03, .03), rs.uniform(-.005, .005), rs.uniform(-.1, .1), 0.0, rs.uniform(-1.5, -0.05))
    m = MG.mono_scene(rs, n, motion, noise=float(rs.choice([0.0, 0.1, 0.6])), out_frac=float(rs.choice([0.0, 0.3, 0.7])))
    kw = dict(height=1.65, pitch=float(rs.choice([0.0, -0.08])), ransac_iters=int(rs.choice([5, 60, 250])),
              inlier_threshold=float(rs.choice([1e-5, 1e-4])), motion_threshold=float(rs.choice([100.0, 30.0])))
    a = B.RefMonoVO(f, cu, cv, **kw); b = B.OracleMonoVO(f, cu, cv, **kw)
    ra = a.process_matches(m); rb = b.process_matches(m)
    if not (ra[0] == rb[0] and ra[1].tobytes() == rb[1].tobytes() and np.array_equal(a.inliers(), b.inliers())):
        bad += 1; print("DIFF", case, n, kw, ra[0], rb[0])
    a.close(); b.close()
print("RESULT", bad)
"""


def test_mono_live_vs_reference(B, have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    root = os.path.dirname(HERE)
    code = _LIVE.format(root=root, tests=HERE, golden=os.path.join(HERE, "golden"), seed=5, cases=40)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT 0" in out.stdout, out.stdout[-3000:]


def test_road_scene_golden(B, synth):
    """stereo + mono VO loops of the oracle on the street scene (depth-dependent disparity and flow)"""
    G.replay_road(load("road_1242x375"), synth, B.OracleStereoVO, B.OracleMonoVO, B.oracle_sampler_seed)
