"""Shared drivers that replay the golden fixtures (tests/golden/*.npz, generated from the real
reference by tests/golden/make_golden.py) through any matcher exposing the CpuMatcher face
(push_back / match / stage / features / ranges / matches / set_intrinsics)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def params_of(g, pi):
    keys = [str(k) for k in g["param_keys"]]
    vals = g[f"p{pi}_params"]
    out = {}
    for k, v in zip(keys, vals):
        out[k] = float(v) if k in ("f", "cu", "cv", "base") else int(v)
    return out


def n_param_sets(g):
    return len([k for k in g.files if k.endswith("_params")])


def replay_small(g, synth, make_matcher, pi, method, check_stages=True):
    """returns number of compared arrays"""
    w, h, nf = int(g["w"]), int(g["h"]), int(g["n_frames"])
    seq = synth.stereo_sequence(int(g["seed"]), w, h, nf, disparity=int(g["disparity"]), ramp=tuple(int(x) for x in g["ramp"]))
    for f, (l, r) in enumerate(seq):
        assert sha(l) + sha(r) == str(g["input_sha"][f]), "synthetic generator drifted from the fixture inputs"
    params = params_of(g, pi)
    m = make_matcher(**params)
    tr = g["tr"] if "tr" in g.files else None
    if "intr" in g.files:
        m.set_intrinsics(*[float(x) for x in g["intr"]])
    ns = 4 if method == 2 else 2
    checked = 0
    for f, (l, r) in enumerate(seq):
        m.push_back(l, r if method else None)
        for s in ("1c1", "1c2", "2c1", "2c2"):
            key = f"p{pi}_f{f}_feat_{s}"
            if key in g.files and (method != 0 or s[0] == "1"):
                got = m.features(s)
                assert same(got, g[key]), f"features {s} frame {f}: got {got.shape} want {g[key].shape}"
                checked += 1
        T = tr if (tr is not None and f >= 2) else None
        ran = m.match(method, T)
        key = f"p{pi}_m{method}_f{f}"
        assert int(ran) == int(g[key + "_ran"]), (key, ran)
        if ran:
            if check_stages:
                for s in range(5):
                    got = m.stage(s)
                    want = g[key + f"_stage{s}"]
                    assert same(got, want), f"{key} stage {s}: got {got.shape} want {want.shape}"
                    checked += 1
                if params["multi_stage"]:
                    assert np.array_equal(m.ranges()[:, :, :ns], g[key + "_ranges"][:, :, :ns]), key + " ranges"
                    checked += 1
            assert same(m.matches(), g[key + "_stage4"]), key + " final"
            checked += 1
    m.close()
    return checked


def replay_hashed(g, synth, make_matcher, frames=None, check_stages=True):
    w, h, nf, method = int(g["w"]), int(g["h"]), int(g["n_frames"]), int(g["method"])
    blur = int(g["blur"])
    if method == 0:
        seq = [(x, None) for x in synth.mono_sequence(int(g["seed"]), w, h, nf, blur=blur)]
    else:
        seq = synth.stereo_sequence(int(g["seed"]), w, h, nf, blur=blur)
    m = make_matcher()
    sets = ("1c1", "1c2") if method == 0 else ("1c1", "1c2", "2c1", "2c2")
    nf = nf if frames is None else min(nf, frames)
    for f in range(nf):
        l, r = seq[f]
        assert sha(l) + (sha(r) if r is not None else "") == str(g["input_sha"][f])
        m.push_back(l, r)
        ran = m.match(method)
        for k, s in enumerate(sets):
            got = m.features(s)
            assert len(got) == int(g["counts"][f][k]), (f, s, len(got), int(g["counts"][f][k]))
            assert sha(got) == str(g["hashes"][f][k]), (f, s)
        if ran and check_stages:
            for s in range(5):
                got = m.stage(s)
                assert len(got) == int(g["counts"][f][len(sets) + s]), (f, "stage", s, len(got))
                assert sha(got) == str(g["hashes"][f][len(sets) + s]), (f, "stage", s)
        fin = m.matches()
        assert len(fin) == int(g["counts"][f][-1]) and sha(fin) == str(g["hashes"][f][-1]), (f, "final")
    m.close()


def replay_vo_sequence(g, synth, make_matcher, n_frames=None):
    """config 2: quad matching with the replayed Tr_delta feedback; final list per frame"""
    w, h = int(g["w"]), int(g["h"])
    nf = int(g["n_frames"]) if n_frames is None else n_frames
    cv = synth.canvas(int(g["seed"]), w, h)
    m = make_matcher()
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    for f in range(nf):
        l, r = synth.stereo_frame(cv, f, w, h)
        assert sha(l) + sha(r) == str(g["input_sha"][f])
        m.push_back(l, r)
        m.match(2, g["tr_in"][f] if bool(g["tr_valid"][f]) else None)
        fin = m.matches()
        assert len(fin) == int(g["counts"][f]), (f, len(fin), int(g["counts"][f]))
        assert sha(fin) == str(g["hashes"][f]), f
    m.close()


def replay_ego_cases(g, make_vo, reset_sampler):
    """ego_cases.npz: egomotion on given match lists; `make_vo(**ego)` returns an object with
    process_matches / inliers / close, `reset_sampler()` puts the RANSAC sampler into the state of
    a fresh process (the fixture was recorded from one, walking the cases in this order)."""
    reset_sampler()
    f, cu, cv, base = [float(x) for x in g["intr"]]
    for ci in range(int(g["n_cases"])):
        m = g[f"c{ci}_matches"]
        it, thr, rw = g[f"c{ci}_ego"]
        vo = make_vo(f, cu, cv, base, ransac_iters=int(it), inlier_threshold=float(thr), reweighting=bool(rw))
        ok, T = vo.process_matches(m)
        assert ok == bool(g[f"c{ci}_ok"][0]), ci
        assert T.tobytes() == g[f"c{ci}_T"][0].tobytes(), (ci, np.abs(T - g[f"c{ci}_T"][0]).max())
        assert np.array_equal(vo.inliers(), g[f"c{ci}_inliers"]), ci
        ok2, T2 = vo.process_matches(m[: max(len(m) // 2, 3)])
        assert ok2 == bool(g[f"c{ci}_ok"][1]), ci
        assert T2.tobytes() == g[f"c{ci}_T"][1].tobytes(), ci
        assert np.array_equal(vo.inliers(), g[f"c{ci}_inliers2"]), ci
        vo.close()


def replay_ego_sequence(g, synth, make_vo, reset_sampler, n_frames=None):
    """*_ego.npz: the live VisualOdometryStereo::process loop; every frame's result flag, Tr_delta,
    bucketed list and inlier set must equal the reference's."""
    reset_sampler()
    w, h = int(g["w"]), int(g["h"])
    nf = int(g["n_frames"]) if n_frames is None else n_frames
    cv = synth.canvas(int(g["seed"]), w, h)
    vo = make_vo(*[float(x) for x in g["intr"]])
    for f in range(nf):
        l, r = synth.stereo_frame(cv, f, w, h)
        assert sha(l) + sha(r) == str(g["input_sha"][f])
        res = vo.process(l, r)
        ok, tout = res[0], res[-1]
        assert ok == bool(g["ok"][f]), f
        b, i = vo.bucketed(), vo.inliers()
        assert len(b) == int(g["n_bucketed"][f]) and sha(b) == str(g["bucketed_sha"][f]), f
        assert len(i) == int(g["n_inliers"][f]) and sha(i) == str(g["inliers_sha"][f]), f
        assert tout.tobytes() == g["tr_out"][f].tobytes(), (f, np.abs(tout - g["tr_out"][f]).max())
    vo.close()


def replay_mono_cases(g, make_vo, reset_sampler):
    """mono_cases.npz: monocular egomotion on given flow-match lists (see replay_ego_cases)"""
    reset_sampler()
    f, cu, cv = [float(x) for x in g["calib"]]
    for ci in range(int(g["n_cases"])):
        m = g[f"c{ci}_matches"]
        height, pitch, it, thr, mot = g[f"c{ci}_mono"]
        vo = make_vo(f, cu, cv, height=float(height), pitch=float(pitch), ransac_iters=int(it),
                     inlier_threshold=float(thr), motion_threshold=float(mot))
        ok, T = vo.process_matches(m)
        assert ok == bool(g[f"c{ci}_ok"][0]), ci
        assert np.array_equal(vo.inliers(), g[f"c{ci}_inliers"]), ci
        assert T.tobytes() == g[f"c{ci}_T"][0].tobytes(), (ci, np.abs(T - g[f"c{ci}_T"][0]).max())
        ok2, T2 = vo.process_matches(m[: max(len(m) // 3, 5)])
        assert ok2 == bool(g[f"c{ci}_ok"][1]), ci
        assert np.array_equal(vo.inliers(), g[f"c{ci}_inliers2"]), ci
        assert T2.tobytes() == g[f"c{ci}_T"][1].tobytes(), ci
        vo.close()


def replay_mono_sequence(g, synth, make_vo, reset_sampler):
    """mono_seq*.npz: VisualOdometryMono::process on images, frame by frame"""
    reset_sampler()
    w, h, nf = int(g["w"]), int(g["h"]), int(g["n_frames"])
    seq = synth.mono_sequence(int(g["seed"]), w, h, nf)
    vo = make_vo(float(g["f"]), w / 2.0, h / 2.0, height=1.65, pitch=-0.08, ransac_iters=300)
    for f in range(nf):
        assert sha(seq[f]) == str(g["input_sha"][f])
        ok, T = vo.process(seq[f])
        assert ok == bool(g["ok"][f]), f
        b, i = vo.bucketed(), vo.inliers()
        assert len(b) == int(g["n_bucketed"][f]) and sha(b) == str(g["bucketed_sha"][f]), f
        assert len(i) == int(g["n_inliers"][f]) and sha(i) == str(g["inliers_sha"][f]), f
        assert T.tobytes() == g["T"][f].tobytes(), f
    vo.close()


def replay_road(g, synth, make_stereo, make_mono, reset_sampler):
    """road_*.npz: street scene with real depth structure; the live stereo VO loop, then the mono VO
    loop, every frame's result flag, Tr_delta, bucketed list and inlier set equal to the reference's"""
    reset_sampler()
    w, h = int(g["w"]), int(g["h"])
    pyr = synth.road_pyramid(int(g["seed"]))
    f, cu, cv, base = [float(x) for x in g["calib"]]
    vo = make_stereo(f, cu, cv, base)
    for k in range(int(g["n_stereo"])):
        l, r = synth.road_stereo_frame(pyr, k, w, h)
        assert sha(l) + sha(r) == str(g["s_in"][k])
        res = vo.process(l, r)
        assert res[0] == bool(g["s_ok"][k]), k
        b, i = vo.bucketed(), vo.inliers()
        assert len(b) == int(g["s_nb"][k]) and sha(b) == str(g["s_hb"][k]), k
        assert len(i) == int(g["s_ni"][k]) and sha(i) == str(g["s_hi"][k]), k
        assert res[-1].tobytes() == g["s_T"][k].tobytes(), k
    vo.close()
    mo = make_mono(f, cu, cv, height=1.65, pitch=0.0)
    for k in range(int(g["n_mono"])):
        img = synth.road_mono_frame(pyr, k, w, h)
        assert sha(img) == str(g["m_in"][k])
        ok, T = mo.process(img)
        assert ok == bool(g["m_ok"][k]), k
        b, i = mo.bucketed(), mo.inliers()
        assert len(b) == int(g["m_nb"][k]) and sha(b) == str(g["m_hb"][k]), k
        assert len(i) == int(g["m_ni"][k]) and sha(i) == str(g["m_hi"][k]), k
        assert T.tobytes() == g["m_T"][k].tobytes(), k
    mo.close()
