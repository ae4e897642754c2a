"""Pins the CPU oracle (oracle/viso_oracle.c) against the REAL reference compiled in
place (oracle/_ref/libvisoref.so, built by `make -C oracle ref` where /root/reference
exists; the built .so travels to the GPU box).  Skipped when that build is absent --
tests/test_golden.py then still pins the oracle through the committed fixtures.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(
    not __import__("oracle.bindings", fromlist=["x"]).have_ref(), reason="oracle/_ref/libvisoref.so not built")


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def _tri_set(t):
    return sorted(tuple(sorted(map(int, r))) for r in t)


@pytest.mark.parametrize("w,h", [(64, 48), (320, 96), (333, 101), (640, 480)])
def test_filters(B, synth, w, h):
    img = B.pad_image(synth.stereo_sequence(7, w, h, 1)[0][0])
    n = img.size
    duo, dvo = B.sobel5x5("oracle", img)
    dur, dvr = B.sobel5x5("ref", img)
    # bytes 0,1 are never written by the reference and the last two read past its temp buffer
    assert np.array_equal(duo.ravel()[2:n - 2], dur.ravel()[2:n - 2])
    assert np.array_equal(dvo.ravel()[2:n - 2], dvr.ravel()[2:n - 2])
    f1o, f1r = B.blob5x5("oracle", img), B.blob5x5("ref", img)
    f2o, f2r = B.checkerboard5x5("oracle", img), B.checkerboard5x5("ref", img)
    assert np.array_equal(f1o[3:h - 3, 3:w - 3], f1r[3:h - 3, 3:w - 3])
    assert np.array_equal(f2o[3:h - 3, 3:w - 3], f2r[3:h - 3, 3:w - 3])
    assert np.array_equal(B.half_image("oracle", img, w), B.half_image("ref", img, w))
    for n_ in (1, 2, 3, 5, 9, 10):
        for tau in (20, 50, 200):
            assert np.array_equal(B.nms("oracle", f1r, f2r, w, n_, tau), B.nms("ref", f1r, f2r, w, n_, tau))


def test_sobel_saturated_and_random(B):
    rng = np.random.default_rng(3)
    for img in (rng.integers(0, 256, (40, 64), dtype=np.uint8), np.kron(rng.integers(0, 2, (10, 16)) * 255, np.ones((4, 4))).astype(np.uint8)):
        img = np.ascontiguousarray(img)
        n = img.size
        a, b = B.sobel5x5("oracle", img), B.sobel5x5("ref", img)
        assert np.array_equal(a[0].ravel()[2:n - 2], b[0].ravel()[2:n - 2])
        assert np.array_equal(a[1].ravel()[2:n - 2], b[1].ravel()[2:n - 2])


@pytest.mark.parametrize("seed", range(6))
def test_delaunay_integer_grids(B, seed):
    """massively co-circular / collinear / duplicated integer point sets: the triangulation
    (not just *a* Delaunay triangulation) must equal Triangle's"""
    rng = np.random.default_rng(seed)
    cases = []
    for n, span in ((4, 3), (5, 4), (12, 4), (50, 8), (200, 12), (500, 40), (3000, 120), (8000, 600)):
        pts = np.stack([rng.integers(0, span * 2, n) * 2, rng.integers(0, span, n) * 2], 1)
        cases.append(pts)
    g = np.stack(np.meshgrid(np.arange(0, 40, 2), np.arange(0, 30, 2)), -1).reshape(-1, 2)
    cases.append(g)                                  # full lattice
    cases.append(g[rng.permutation(len(g))])         # same, shuffled input order
    cases.append(np.stack([np.arange(20) * 2, np.full(20, 6)], 1))  # all collinear -> no triangles
    cases.append(np.concatenate([g[:50], g[:50]]))   # every point twice
    for pts in cases:
        a = B.delaunay("oracle", pts.astype(np.float32))
        b = B.delaunay("ref", pts.astype(np.float32))
        assert len(a) == len(b)
        assert _tri_set(a) == _tri_set(b)


PARAM_SETS = [
    dict(),
    dict(half_resolution=0),
    dict(multi_stage=0),
    dict(refinement=0),
    dict(refinement=2),
    dict(refinement=2, half_resolution=0),
    dict(nms_n=2, nms_tau=30, match_binsize=32, match_radius=120, match_disp_tolerance=1),
    dict(nms_n=5, outlier_flow_tolerance=3, outlier_disp_tolerance=3),
]


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("pi", range(len(PARAM_SETS)))
def test_matcher_stages(B, synth, method, pi):
    params = PARAM_SETS[pi]
    w, h = (416, 160)
    seq = synth.stereo_sequence(100 + pi, w, h, 4, disparity=12, ramp=(1, 16))
    mo, mr = B.CpuMatcher("oracle", **params), B.CpuMatcher("ref", **params)
    ns = 4 if method == 2 else 2
    for f, (l, r) in enumerate(seq):
        replace = (f == 2)  # exercises the overwrite-current branch of pushBack
        for m in (mo, mr):
            m.push_back(l, r if method else None, replace=replace)
        for s in ("1p1", "1p2", "1c1", "1c2", "2c1", "2c2"):
            assert _same(mo.features(s), mr.features(s)), (f, s)
        ro, rr = mo.match(method), mr.match(method)
        assert ro == rr
        if ro:
            for s in range(5):
                assert _same(mo.stage(s), mr.stage(s)), (f, "stage", s)
            if mo.p["multi_stage"]:
                assert np.array_equal(mo.ranges()[:, :, :ns], mr.ranges()[:, :, :ns])
        assert _same(mo.matches(), mr.matches())
    assert len(mo.matches()) > 20


def test_quad_with_tr_delta(B, synth):
    """Tr_delta prediction path (viso/matcher.cpp:1114-1136, :948-953): double sqrt costs"""
    w, h = 480, 200
    f_, cu, cv, base = 400.0, 240.5, 99.25, 0.54
    seq = synth.stereo_sequence(5, w, h, 4, disparity=16, ramp=(1, 20))
    Tr = np.eye(4)
    Tr[0, 3], Tr[2, 3] = 0.011, -0.35
    Tr[0, 2], Tr[2, 0] = 0.004, -0.004
    for params in (dict(), dict(multi_stage=0), dict(half_resolution=0)):
        mo, mr = B.CpuMatcher("oracle", **params), B.CpuMatcher("ref", **params)
        for m in (mo, mr):
            m.set_intrinsics(f_, cu, cv, base)
        for f, (l, r) in enumerate(seq):
            for m in (mo, mr):
                m.push_back(l, r)
            ro, rr = mo.match(2, Tr if f > 1 else None), mr.match(2, Tr if f > 1 else None)
            assert ro == rr
            for s in range(5):
                assert _same(mo.stage(s), mr.stage(s)), (params, f, s)
        assert len(mo.matches()) > 20


def test_bucketing_and_gain(B, synth):
    import ctypes
    libc = ctypes.CDLL(None)
    w, h = 416, 160
    seq = synth.stereo_sequence(9, w, h, 3)
    res = []
    for kind in ("oracle", "ref"):
        m = B.CpuMatcher(kind)
        libc.srand(0)  # VisualOdometry::VisualOdometry, viso/viso.cpp:35
        for l, r in seq:
            m.push_back(l, r)
            m.match(2)
            m.bucket(2, 50.0, 50.0)
        g = m.gain(np.arange(0, len(m.matches()), 2))
        res.append((m.matches(), g))
    assert _same(res[0][0], res[1][0])
    assert res[0][1] == res[1][1]
    assert 0 < len(res[0][0]) < 200


def test_error_paths(B, synth):
    l, r = synth.stereo_sequence(1, 128, 64, 1)[0]
    for kind in ("oracle", "ref"):
        m = B.CpuMatcher(kind)
        assert not m.match(2)          # nothing pushed: silent return
        m.push_back(l, r)
        assert not m.match(2)          # only one frame
        assert not m.match(0)
        assert len(m.matches()) == 0
    mo = B.CpuMatcher("oracle")
    assert mo.L.vo_push_back(mo.h, None, None, 128, 64, 128, 0) == -1
    assert mo.L.vo_push_back(mo.h, l.ctypes.data_as(B._p_u8), None, 128, 64, 100, 0) == -1


def test_remove_outliers_on_the_chain_test_lists(B):
    """Matcher::removeOutliers (viso/matcher.cpp:1207-1377) on the match lists the GPU suite feeds the device chain
    (tools/dc2_check.py make_list: random and grid-shaped point sets with shared pixels, list lengths around the chain's
    structural boundaries): the oracle's survivors against the reference's own.  The GPU tests then compare the device
    chain with the oracle only (the reference's library is not loaded beside the HIP runtime)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dc2_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dc2_check.py"))
    dc2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc2)
    n_checked = 0
    for n in (0, 3, 4, 5, 17, 100, 480, 481, 961, 2000, 4500, 7400, 9000, 12289):
        for method in (0, 1, 2):
            for grid in (False, True):
                lst = dc2.make_list(n, grid=grid)
                a, b = B.remove_outliers("oracle", lst, method), B.remove_outliers("ref", lst, method)
                assert len(a) == len(b) and a.tobytes() == b.tobytes(), (n, method, grid, len(a), len(b))
                n_checked += len(a)
    assert n_checked > 100000


def test_reference_crashes_on_a_one_pixel_list(B, have_ref):
    """A defect of the reference, pinned so that nobody goes looking for it in the GPU stack again: removeOutliers on more than
    three matches that all lie on ONE pixel never returns - Triangle's divconqrecurse (viso/triangle.cpp:5966-6092) splits
    the single distinct vertex into 0 + 1 and calls itself on the 1 until the stack overflows (SIGSEGV).  Run in a process
    of its own; the oracle's answer for the same list is "nothing survives"."""
    if not have_ref:
        pytest.skip("compiled reference not available")
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib.util, sys
        sys.path.insert(0, %r)
        from oracle import bindings as B
        spec = importlib.util.spec_from_file_location("dc2_check", %r)
        dc2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(dc2)
        lst = dc2.make_list(60, dup=0)
        lst["u1c"], lst["v1c"] = 100, 50
        print("oracle", len(B.remove_outliers("oracle", lst, 2)), flush=True)
        B.remove_outliers("ref", lst, 2)
        print("reference returned", flush=True)
    """) % (root, os.path.join(root, "tools", "dc2_check.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "oracle 0" in r.stdout
    assert r.returncode == -11 and "reference returned" not in r.stdout, (r.returncode, r.stdout, r.stderr[-400:])
