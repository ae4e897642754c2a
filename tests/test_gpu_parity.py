"""GPU parity suite (run with -m gpu on an MI355X): the HIP path, called through the C-ABI
(libvisomatch.so), must be bit-identical to the CPU oracle on seeded inputs and to the committed
golden vectors of the real reference, stage by stage."""
import os

import numpy as np
import pytest

import golden_util as G
from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vm():
    m = pkg("visomatch")
    m.lib()  # raises if the HIP library is missing: no silent fallback
    return m


def _gpu(vm):
    return lambda **p: vm.Matcher(stage_capture=True, **p)


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("w,h,half", [(320, 96, 1), (333, 101, 1), (640, 480, 1), (1242, 375, 1), (304, 128, 0), (64, 48, 0)])
def test_filters_vs_oracle(vm, B, synth, w, h, half, fused):
    """whole planes; fused = 1: the responses are a side output of the fused filter + suppression tiles (k_feat_dense)"""
    l, r = synth.stereo_sequence(21, w, h, 1)[0]
    m = vm.Matcher(half_resolution=half, options={"fused_features": fused, "filter_planes": 1})
    assert m.push_back(l, r) == 0
    img = B.pad_image(l)
    mimg = B.half_image("oracle", img, w) if half else img
    du, dv = B.sobel5x5("oracle", mimg)
    gdu, gdv = m.gradients(2, False)
    assert np.array_equal(gdu, du.ravel()) and np.array_equal(gdv, dv.ravel())
    if half:
        duf, dvf = B.sobel5x5("oracle", img)
        gduf, gdvf = m.gradients(2, True)
        assert np.array_equal(gduf, duf.ravel()) and np.array_equal(gdvf, dvf.ravel())
    f1, f2 = m.filter_responses()
    assert np.array_equal(f1, B.blob5x5("oracle", mimg).ravel())
    assert np.array_equal(f2, B.checkerboard5x5("oracle", mimg).ravel())
    # right image gradients too
    rimg = B.pad_image(r)
    rm = B.half_image("oracle", rimg, w) if half else rimg
    du2, _ = B.sobel5x5("oracle", rm)
    assert np.array_equal(m.gradients(3, False)[0], du2.ravel())
    m.close()
    if fused:  # without the option the fused tiles keep f1 / f2 in LDS: nothing to read back
        m = vm.Matcher(half_resolution=half)
        assert m.push_back(l, r) == 0 and m.filter_responses() == (None, None)
        m.close()


@pytest.mark.parametrize("method", [2, 0, 1])
@pytest.mark.parametrize("pi", range(6))
def test_golden_small(vm, synth, pi, method):
    assert G.replay_small(G.load("small_304x128"), synth, _gpu(vm), pi, method) > 5


@pytest.mark.parametrize("pi", range(2))
def test_golden_small_tr_delta(vm, synth, pi):
    assert G.replay_small(G.load("small_tr_352x160"), synth, _gpu(vm), pi, 2) > 5


def test_golden_cfgA(vm, synth):
    G.replay_hashed(G.load("cfgA_1242x375_quad"), synth, _gpu(vm))


def test_golden_cfg3_mono(vm, synth):
    G.replay_hashed(G.load("cfg3_640x480_mono"), synth, _gpu(vm))


def test_golden_cfg5_highres(vm, synth):
    G.replay_hashed(G.load("cfg5_2048x1024_quad"), synth, _gpu(vm))


@pytest.mark.parametrize("full", ["0", "1"])
def test_golden_cfg5_highres_lookahead_gpu_final_stage(vm, synth, monkeypatch, full):
    """2048x1024 (33 k matches per pair: more points than the packed record form and the 16-bit paths take) through
    the look-ahead API with the final stage forced onto the GPU share: the reference's final lists"""
    g = G.load("cfg5_2048x1024_quad")
    w, h, nf, method = int(g["w"]), int(g["h"]), int(g["n_frames"]), int(g["method"])
    seq = synth.stereo_sequence(int(g["seed"]), w, h, nf, blur=int(g["blur"]))
    opt = {"dc_gpu": 1, "dc_full": int(full)}
    m = vm.Matcher(options=opt)
    got = m.run_sequence(np.stack([l for l, _ in seq]), np.stack([r for _, r in seq]), method)
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f][-1]) and G.sha(got[f]) == str(g["hashes"][f][-1]), f
    m.close()


def test_golden_cfg5_as_specified_20k(vm, synth):
    """config 5 as BASELINE.json words it (~20 k dense features per 2048x1024 image; blur radius 29, see make_golden.py):
    six frames through the per-frame API, features and every match stage against the reference's hashes"""
    G.replay_hashed(G.load("cfg5_2048x1024_quad_20k"), synth, _gpu(vm))


@pytest.mark.parametrize("form", ["host-shared, final stage all on the GPU", "host-shared, final stage shared", "GPU-resident"])
def test_golden_cfg5_as_specified_20k_lookahead(vm, synth, monkeypatch, form):
    """... and through the look-ahead API (chunks of 3 frames) with the exact Delaunay stage on the GPU in every form:
    16.6 k matches per pair = more points than the LDS-whole merge levels and the 16-bit kd lists take"""
    import torch
    g = G.load("cfg5_2048x1024_quad_20k")
    w, h, nf, method = int(g["w"]), int(g["h"]), int(g["n_frames"]), int(g["method"])
    seq = synth.stereo_sequence(int(g["seed"]), w, h, nf, blur=int(g["blur"]))
    monkeypatch.setenv("VSM_SEQ_CHUNK", "3")
    monkeypatch.setenv("VSM_SEQ_V2", "1" if form == "GPU-resident" else "0")
    opt = {"dc_gpu": 1, "dc_full": 1 if "all on the GPU" in form else 0}
    m = vm.Matcher(options=opt)
    left = torch.from_numpy(np.stack([l for l, _ in seq])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in seq])).cuda()
    got = m.run_sequence(left, right, method)
    assert m.sequence_path() == (2 if form == "GPU-resident" else 1)
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f][-1]) and G.sha(got[f]) == str(g["hashes"][f][-1]), f
    m.close()


def test_golden_cfg2_sequence_feedback(vm, synth):
    G.replay_vo_sequence(G.load("cfg2_seq200_tr"), synth, _gpu(vm), n_frames=40)


@pytest.mark.parametrize("method", [0, 1, 2])
def test_vs_oracle_replace_and_params(vm, B, synth, method):
    """seeded inputs through oracle and GPU side by side, incl. replace=True and odd sizes"""
    for params in (dict(), dict(nms_n=5, outlier_flow_tolerance=3), dict(match_binsize=40, match_radius=150)):
        seq = synth.stereo_sequence(77, 417, 163, 4, disparity=10, ramp=(1, 12))
        g, c = vm.Matcher(stage_capture=True, **params), B.CpuMatcher("oracle", **params)
        for f, (l, r) in enumerate(seq):
            rep = f == 2
            g.push_back(l, r if method else None, replace=rep)
            c.push_back(l, r if method else None, replace=rep)
            for s in ("1p1", "1p2", "1c1", "1c2", "2c2"):
                assert _same(g.features(s), c.features(s)), (params, f, s)
            assert g.match(method) == c.match(method)
            for s in range(5):
                assert _same(g.stage(s), c.stage(s)), (params, f, s)
        g.close()


def test_device_resident_inputs(vm, B, synth):
    import torch
    seq = synth.stereo_sequence(5, 500, 200, 3)
    g, c = vm.Matcher(), B.CpuMatcher("oracle")
    for l, r in seq:
        tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
        assert g.push_back(tl, tr) == 0
        c.push_back(l, r)
        g.match(2)
        c.match(2)
        assert _same(g.matches(), c.matches())
    assert len(g.matches()) > 100


def test_bucketing_gain_and_errors(vm, B, synth):
    import ctypes
    libc = ctypes.CDLL(None)
    seq = synth.stereo_sequence(9, 416, 160, 3)
    res = []
    for make in (lambda: vm.Matcher(), lambda: B.CpuMatcher("oracle")):
        m = make()
        libc.srand(0)
        assert not m.match(2)
        for l, r in seq:
            m.push_back(l, r)
            m.match(2)
            m.bucket(2, 50.0, 50.0)
        res.append((m.matches(), m.gain(np.arange(0, len(m.matches()), 2))))
    assert _same(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    m = vm.Matcher()
    L = vm.lib()
    assert L.vsm_push_back(m.h, None, None, 128, 64, 128, 0) == m.EDIMS
    assert m.match_features(2) == m.ENOTREADY
    l, r = seq[0]
    m.push_back(l, r)
    assert m.match_features(2) == m.ENOTREADY and m.match_features(0) == m.ENOTREADY
    assert len(m.get_matches()) == 0


def test_full_size_properties(vm, synth):
    """size-independent properties at BASELINE's full size: index validity, ordering, closure"""
    seq = synth.stereo_sequence(1234, 1242, 375, 3)
    m = vm.Matcher(stage_capture=True)
    for l, r in seq:
        m.push_back(l, r)
        m.match(2)
    fin, raw = m.matches(), m.stage(2)
    n1p, n1c = len(m.features("1p2")), len(m.features("1c2"))
    assert len(fin) > 5000
    assert np.all(np.diff(raw["i1p"]) > 0)            # emitted in ascending query order
    assert np.all(np.diff(fin["i1p"]) > 0)            # outlier removal preserves order
    assert fin["i1p"].max() < n1p and fin["i1c"].max() < n1c
    assert np.all(raw["u1p"] >= raw["u2p"]) and np.all(raw["u1c"] >= raw["u2c"])
    f1p = m.features("1p2")
    assert np.array_equal(f1p[raw["i1p"], 0], raw["u1p"].astype(np.int32))
    # idempotence: matching again without a push gives the same list
    m.match(2)
    assert _same(m.matches(), fin)


@pytest.mark.parametrize("binary", ["vo_dropin", "vo_native"])
def test_reference_vo_runs_on_dropin_matcher(synth, tmp_path, binary):
    """tools/dropin/_build/vo_dropin = the reference's unmodified VisualOdometryStereo sources
    compiled against include/matcher.h + libvisomatch.so (built where /root/reference exists);
    vo_native = the same driver compiled against include/viso_stereo.h (the C-ABI's own
    VisualOdometryStereo; only the reference's Matrix class is compiled in).
    Their per-frame Tr_delta must equal what the all-reference build produced (golden cfg2)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "dropin", "_build", binary)
    if not os.path.exists(exe):
        pytest.skip("drop-in binary not built (needs the reference sources at build time)")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 30
    cv = synth.canvas(int(g["seed"]), w, h)
    raw = tmp_path / "frames.raw"
    with open(raw, "wb") as f:
        f.write(np.array([w, h, nf], dtype=np.int32).tobytes())
        for i in range(nf):
            l, r = synth.stereo_frame(cv, i, w, h)
            f.write(l.tobytes())
            f.write(r.tobytes())
    out = tmp_path / "out.bin"
    intr = [repr(float(x)) for x in g["intr"]]
    subprocess.check_call([exe, str(raw), str(out)] + intr, timeout=300)
    rec = np.fromfile(out, dtype=np.float64).reshape(nf, 18)
    for i in range(nf - 1):
        assert bool(rec[i, 0]) == bool(g["vo_ok"][i])
        # Tr_delta after frame i is what matchFeatures receives at frame i+1
        assert np.array_equal(rec[i, 2:].reshape(4, 4), g["tr_in"][i + 1]), i


@pytest.mark.parametrize("chunk", ["1", "3", "50"])
@pytest.mark.parametrize("method", [2, 0, 1])
@pytest.mark.parametrize("final_stage", ["as it comes", "shared with the GPU", "all on the GPU", "GPU-resident form"])
def test_sequence_api_equals_frame_by_frame(vm, B, synth, monkeypatch, method, chunk, final_stage):
    """look-ahead API == pushBack+matchFeatures per frame (oracle), across chunk boundaries; with the final stage
    (exact Delaunay support test) where the chunk size puts it - the host for these short chunks - and forced
    through both forms of the GPU share, for flow, stereo and quad matching; and through the GPU-resident form of the
    whole path (lists stay in HBM, removeOutliers + prior statistics as device kernels)"""
    monkeypatch.setenv("VSM_SEQ_CHUNK", chunk)
    monkeypatch.setenv("VSM_SEQ_V2", "1" if final_stage == "GPU-resident form" else "0")
    opt = {}
    if final_stage not in ("as it comes", "GPU-resident form"):
        opt = {"dc_gpu": 1, "dc_full": 1 if final_stage == "all on the GPU" else 0}
    seq = synth.stereo_sequence(31, 417, 163, 7, disparity=10, ramp=(1, 12))
    left = np.stack([l for l, _ in seq])
    right = np.stack([r for _, r in seq])
    g = vm.Matcher(options=opt)
    got = g.run_sequence(left, right, method)
    assert g.sequence_path() == (2 if final_stage == "GPU-resident form" else 1)
    c = B.CpuMatcher("oracle")
    for f, (l, r) in enumerate(seq):
        c.push_back(l, r)
        c.match(method)
        assert _same(got[f], c.matches()), (method, chunk, f, len(got[f]), len(c.matches()))
    assert len(got[-1]) > 100
    g.close()


@pytest.mark.parametrize("form", ["host-shared", "GPU-resident"])
@pytest.mark.parametrize("chunk", ["50", "16", "17", "7"])
def test_sequence_api_golden_feedback_and_device_inputs(vm, synth, monkeypatch, chunk, form):
    """config 2 through the look-ahead API: 60 frames resident in HBM, replayed Tr_delta; chunk 16 / 17 make
    it a sequence of four chunks (three frame banks, the pair banks and the slab banks all come round), chunk 7 one of nine"""
    import torch
    monkeypatch.setenv("VSM_SEQ_CHUNK", chunk)
    monkeypatch.setenv("VSM_SEQ_V2", "1" if form == "GPU-resident" else "0")
    opt = {"dc_gpu": 1}   # (chunks with fewer pairs than host threads would stay on the host)
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 60
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    m = vm.Matcher(options=opt)
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), f
    m.close()


@pytest.mark.parametrize("env", [{}, {"VSM_SEQ_SERIAL": "1"}, {"VSM_SEQ_EARLY_EXPORT": "0"},
                                 {"opt:seq_null_stream": "0"}, {"opt:seq_null_stream": "0", "VSM_SEQ_GPU_SORTS": "50", "VSM_SEQ_CHUNK": "5"},
                                 {"opt:fused_features": "0"}, {"opt:feat_order": "0"}, {"opt:fused_features": "0", "opt:feat_order": "0", "opt:front": "0"},
                                 {"VSM_SEQ_EARLY_EXPORT": "1", "VSM_HOST_THREADS": "2"}, {"VSM_SEQ_DC_STREAMS": "1"},
                                 {"VSM_SEQ_DC_STREAMS": "4", "VSM_SEQ_CHUNK": "5"}, {"VSM_SEQ_CHUNK": "2", "VSM_SEQ_EARLY_EXPORT": "0"},
                                 {"VSM_SEQ_GPU_SORTS": "100"}, {"VSM_SEQ_GPU_SORTS": "40", "VSM_HOST_THREADS": "2"},
                                 {"VSM_SEQ_GPU_SORTS": "50", "VSM_SEQ_CHUNK": "5"}, {"VSM_SEQ_GPU_SORTS": "50", "VSM_SEQ_CHUNK": "4", "opt:seq_p2_first": "1"},
                                 {"VSM_HOST_THREADS": "2", "VSM_SEQ_CHUNK": "19"}, {"VSM_HOST_THREADS": "1", "VSM_SEQ_CHUNK": "20"},
                                 {"VSM_HOST_THREADS": "3", "VSM_SEQ_CHUNK": "13"},
                                 {"opt:front": "0"}, {"opt:seq_p2_first": "1"}, {"opt:seq_p2_first": "0", "VSM_HOST_THREADS": "6"},
                                 {"opt:seq_p2_first": "1", "opt:seq_first_chunk": "4"}, {"opt:seq_keys_dma": "0", "opt:seq_export_budget": "0"},
                                 {"opt:seq_ties1_null": "0", "opt:seq_last_first": "0", "opt:seq_export_budget": "5"},
                                 {"opt:match_heads": "1"}, {"opt:match_heads": "1", "opt:feat_order": "0", "VSM_SEQ_CHUNK": "7"},
                                 {"opt:seq_keys_dma": "1", "opt:seq_keys_pieces": "1"}, {"opt:seq_keys_dma": "2", "opt:seq_keys_pieces": "3", "VSM_SEQ_CHUNK": "19"},
                                 {"opt:seq_keys_dma": "0"}, {"opt:seq_block_after_p2": "1"}, {"opt:seq_block_after_p2": "1", "VSM_SEQ_CHUNK": "19"},
                                 {"VSM_POLL_SPIN": "1", "VSM_HOST_THREADS": "3"}, {"VSM_POLL_SPIN": "0"},
                                 {"opt:seq_ties1_host": "0"}, {"opt:seq_warm_gaps": "0"}, {"opt:seq_warm_gaps": "1", "VSM_HOST_THREADS": "12", "VSM_SEQ_CHUNK": "9"}, {"opt:seq_ties1_host": "1", "VSM_HOST_THREADS": "2", "VSM_SEQ_CHUNK": "7"},
                                 {"opt:seq_ties1_host": "1", "opt:seq_p2_first": "1", "VSM_SEQ_CHUNK": "5"}])
def test_gpu_resident_form_switches(vm, synth, monkeypatch, env):
    """The GPU-resident look-ahead form under its switches - nothing overlapping (the bench's `alone` pass), both ways of
    result delivery at both ends of the pool size, one / four chain streams (eight chunks of five, twenty of two: every bank
    comes round), the vertex sorts on the device (one launch for the chunks that wait for it, also where a slab comes round
    before the call's last head; two-chunk calls of ranks with few host threads: a launch per chunk, the pool sizes' own shares), the unfused front end, the
    separate filter / suppression / record / bin kernels instead of the fused tiles, a fifth stream of the library's own instead of the null stream, the scheduling choices of DESIGN_HISTORY.md 6c either way,
    the second pass on per-bin head records (k_feat_heads; measured slower, off by default), the keys' copy on either stream and
    in one / three / four pieces, a chunk's block kernel behind the next chunk's second pass, the poller sleeping or yielding: always
    the reference's lists, and always this form (it must not quietly hand the run to the other one)."""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "1")
    monkeypatch.setenv("VSM_SEQ_CHUNK", "10")
    for k, v in env.items():
        if not k.startswith("opt:"):
            monkeypatch.setenv(k, v)
    opt = {k[4:]: int(v) for k, v in env.items() if k.startswith("opt:")}   # (switches vsm_set_option takes but the environment does not)
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 38
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    m = vm.Matcher(options=opt)
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
    assert m.sequence_path() == 2, env
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (env, f)
    m.close()


@pytest.mark.parametrize("pinned", [0, 1])
def test_lookahead_host_inputs_short_last_chunk_and_page_locked_memory(vm, synth, monkeypatch, pinned):
    """vsm_sequence_run fed from host memory (what Matcher::pushBack takes, viso/matcher.cpp:95-181): chunks of 25 + 50 + 25 + a
    short last one of 20 frames (the call's end waits for what follows the last frames' arrival); pinned = 1: the caller has
    page-locked its arrays (vsm_host_register) and says so (option seq_host_pinned) - the pieces leave straight from its memory,
    row strides included.  The reference's lists either way."""
    monkeypatch.setenv("VSM_SEQ_V2", "1")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 120
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = np.ascontiguousarray(np.stack([l for l, _ in fr]))
    right = np.ascontiguousarray(np.stack([r for _, r in fr]))
    m = vm.Matcher(options={"seq_chunk": 50})
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    if pinned:
        assert vm.host_register(left) and vm.host_register(right)
        m.set_option("seq_host_pinned", 1)
    try:
        for _ in range(2):   # (the second call reuses streams, events and the upload buffer)
            got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
            assert m.sequence_path() == 2
            for f in range(nf):
                assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (pinned, f)
    finally:
        m.close()
        if pinned:
            vm.host_unregister(left)
            vm.host_unregister(right)


@pytest.mark.parametrize("inorder", [0, 1])
def test_lookahead_host_inputs_orders(vm, synth, monkeypatch, inorder):
    """host-fed vsm_sequence_run: chunk by chunk in the order of arrival (default; two chain streams, pass-1 sorts on the main
    stream) and the run-ahead order of resident input (option seq_host_inorder = 0), the library's own chunk plan
    (40 + 80 + 60 + 20), after a call from HBM in the same handle (the third chain stream is given back and comes again)"""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "1")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 200
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = np.ascontiguousarray(np.stack([l for l, _ in fr]))
    right = np.ascontiguousarray(np.stack([r for _, r in fr]))
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    m = vm.Matcher(options={"seq_host_inorder": inorder})
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    try:
        for src in ((dl, dr), (left, right), (dl, dr), (left, right)):
            got = m.run_sequence(src[0], src[1], 2, g["tr_in"][:nf], g["tr_valid"][:nf])
            assert m.sequence_path() == 2
            for f in range(nf):
                assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (inorder, f)
    finally:
        m.close()


def test_device_blocks_wait_for_the_next_handle(vm, synth, monkeypatch):
    """a closed handle's large device blocks go to the process-wide cache, not back to the driver (whose background clear of
    released VRAM slows every call for 44 ms per GB); the next handle takes them - as they are, so nothing may
    rely on fresh memory being zero: same lists as the reference's from re-used blocks, for another chunk size and another
    image size as well; vsm_device_pool_trim() hands them back"""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "1")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 60
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    dl = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    dr = torch.from_numpy(np.stack([r for _, r in fr])).cuda()

    def run(chunk):
        m = vm.Matcher(options={"seq_chunk": chunk})
        m.set_intrinsics(*[float(x) for x in g["intr"]])
        got = m.run_sequence(dl, dr, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
        assert m.sequence_path() == 2
        for f in range(nf):
            assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (chunk, f)
        m.close()

    vm.device_pool_trim()
    assert vm.device_pool_stats()[0] == 0
    run(30)
    blocks, nbytes, _ = vm.device_pool_stats()
    assert blocks >= 2 and nbytes > 256 << 20     # (the context's arena and the chains' slabs at least)
    run(30)
    assert vm.device_pool_stats()[:2] == (blocks, nbytes)   # (everything came out of the cache and went back)
    run(20)                                      # a smaller context in the larger one's arena
    # another image size: per-frame calls of a small handle between two look-ahead handles
    m = vm.Matcher()
    a = torch.from_numpy(np.ascontiguousarray(fr[0][0][:240, :320])).cuda()
    b = torch.from_numpy(np.ascontiguousarray(fr[1][0][:240, :320])).cuda()
    m.push_back(a, None)
    m.push_back(b, None)
    m.match_features(0, None)
    m.close()
    run(30)
    vm.device_pool_trim()
    assert vm.device_pool_stats()[:2] == (0, 0)


def test_host_threads_near_the_gpu(vm):
    """vsm_local_cpus(): the CPUs of the device's NUMA node the library keeps its own threads on - a subset of what the
    process may use, or nothing (one node, or a process already confined); the caller's thread is left alone"""
    import os
    before = os.sched_getaffinity(0)
    m = vm.Matcher()
    cpus = vm.local_cpus()
    assert os.sched_getaffinity(0) == before
    assert set(cpus) <= before and (not cpus or 2 <= len(cpus) < len(before))
    m.close()


def test_lookahead_as_shipped_200_frames(vm, synth, monkeypatch):
    """What bench.py times, as it ships: config 2 (200 frames 1242x375 resident in HBM, replayed Tr_delta) through
    vsm_sequence_run with NO VSM_* variable set - the GPU-resident form, chunks of 110 (80 / 50 with fewer than ten / six host
    threads), three side streams, the default
    host pool - every frame's final list against the reference's hash, twice on one handle (banks and slabs are reused)."""
    import os
    import torch
    for k in [k for k in os.environ if k.startswith("VSM_")]:
        monkeypatch.delenv(k)
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 200
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    m = vm.Matcher()
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    for rep in range(2):
        got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
        assert m.sequence_path() == 2
        assert int(m.sequence_timings()["chunk"]) in (110, 100, 80, 50)   # (by the box's CPU share)
        for f in range(nf):
            assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (rep, f)
    m.close()


@pytest.mark.parametrize("form", ["host-shared", "GPU-resident"])
@pytest.mark.parametrize("method", [0, 1, 2])
def test_sequence_api_single_stage(vm, B, synth, monkeypatch, method, form):
    """multi_stage = 0 (one matching pass, no prior statistics: viso/matcher.cpp:217-221) through the look-ahead API in both
    forms, flow / stereo / quad: the GPU-resident form then has no first-pass chain (and no first-pass slab) at all"""
    monkeypatch.setenv("VSM_SEQ_CHUNK", "3")
    monkeypatch.setenv("VSM_SEQ_V2", "1" if form == "GPU-resident" else "0")
    opt = {"dc_gpu": 1}
    seq = synth.stereo_sequence(33, 417, 163, 7, disparity=10, ramp=(1, 12))
    left = np.stack([l for l, _ in seq])
    right = np.stack([r for _, r in seq])
    g = vm.Matcher(multi_stage=0, options=opt)
    got = g.run_sequence(left, right, method)   # (stereo input also for flow matching: mono input goes frame by frame)
    assert g.sequence_path() == (2 if form == "GPU-resident" else 1)
    c = B.CpuMatcher("oracle", multi_stage=0)
    for f, (l, r) in enumerate(seq):
        c.push_back(l, r)
        c.match(method)
        assert _same(got[f], c.matches()), (method, f, len(got[f]), len(c.matches()))
    assert len(got[-1]) > 100
    g.close()


@pytest.mark.parametrize("device_inputs", [False, True])
@pytest.mark.parametrize("chunk", ["4", "5", "50"])
def test_mono_flow_lookahead(vm, B, synth, monkeypatch, chunk, device_inputs):
    """mono input (Matcher::pushBack(I1, dims, replace) + matchFeatures(0), viso/matcher.cpp:1006-1041) through the batched
    GPU-resident look-ahead form: 13 frames in chunks of 4 (odd and even chunk lengths, four banks come round), 5 and one
    chunk, host and device inputs, against the oracle frame by frame"""
    import torch
    monkeypatch.setenv("VSM_SEQ_CHUNK", chunk)
    seq = synth.stereo_sequence(41, 417, 163, 13, disparity=10, ramp=(1, 12))
    left = np.stack([l for l, _ in seq])
    g = vm.Matcher()
    got = g.run_sequence(torch.from_numpy(left).cuda() if device_inputs else left, None, 0)
    assert g.sequence_path() == 2
    c = B.CpuMatcher("oracle")
    for f, (l, _) in enumerate(seq):
        c.push_back(l, None)
        c.match(0)
        assert _same(got[f], c.matches()), (chunk, f, len(got[f]), len(c.matches()))
    assert len(got[-1]) > 100
    g.close()


def test_golden_cfg3_mono_lookahead(vm, synth):
    """config 3 (640x480 mono, flow matching) through the look-ahead API: the reference's final lists"""
    g = G.load("cfg3_640x480_mono")
    w, h, nf, method = int(g["w"]), int(g["h"]), int(g["n_frames"]), int(g["method"])
    assert method == 0
    seq = synth.mono_sequence(int(g["seed"]), w, h, nf, blur=int(g["blur"]))
    m = vm.Matcher()
    got = m.run_sequence(np.stack(seq), None, 0)
    assert m.sequence_path() == 2
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f][-1]) and G.sha(got[f]) == str(g["hashes"][f][-1]), f
    m.close()


def test_sequence_forms_alternate_on_one_handle(vm, B, synth):
    """a stereo run, then a mono run, then stereo again on ONE handle (the frame banks change their numbering between
    them): the getters always answer for the last run"""
    seq = synth.stereo_sequence(35, 417, 163, 5, disparity=10, ramp=(1, 12))
    left = np.stack([l for l, _ in seq])
    right = np.stack([r for _, r in seq])
    g = vm.Matcher()
    ref = {}
    for method, rgt in ((2, True), (0, False)):
        c = B.CpuMatcher("oracle")
        ref[method] = []
        for l, r in seq:
            c.push_back(l, r if rgt else None)
            c.match(method)
            ref[method].append(c.matches())
    for method, rgt in ((2, True), (0, False), (2, True)):
        got = g.run_sequence(left, right if rgt else None, method)
        assert g.sequence_path() == 2
        for f in range(len(seq)):
            assert _same(got[f], ref[method][f]), (method, f)
    g.close()


@pytest.mark.parametrize("opt", [{"dc_gpu": 0}, {"dc_gpu": 1, "dc_full": 1}, {"dc_gpu": 1, "dc_full": 0}, {"dc_gpu": 1}, {}])
def test_sequence_api_final_stage_variants(vm, synth, monkeypatch, opt):
    """the ways the exact Delaunay of the host-shared look-ahead form can be shared between host and GPU (host only;
    everything after the sort on the GPU; block sub-trees on the GPU and the merges above them on the host; forced or
    left to the chunk size) all give the reference's lists.  (Round 1's further sub-variants of this form are compile-time
    choices of csrc/vsm_api.cpp now.)"""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "0")
    env = opt
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 24
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    monkeypatch.setenv("VSM_SEQ_CHUNK", "10")
    m = vm.Matcher(options=opt)
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), (env, f)
    m.close()


def test_lookahead_calls_complete_under_pool_races(vm, synth):
    """120 look-ahead calls back to back with chunk sizes that shift what the pool's tasks overlap with (the chain's completion
    is seen by a pool task that asks an event; a chunk's state once went 1 -> 3 -> 2 when that task overtook the thread that
    had submitted it, and the call then waited for ever): every call returns, the last call's lists are the reference's"""
    import torch
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 60
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    for chunk in (27, 15, 9):
        m = vm.Matcher(options={"seq_chunk": chunk})
        m.set_intrinsics(*[float(x) for x in g["intr"]])
        for _ in range(40):
            m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf], fetch=False)
        assert m.sequence_path() == 2
        for f in range(nf):
            got = m.sequence_matches(f)
            assert len(got) == int(g["counts"][f]) and G.sha(got) == str(g["hashes"][f]), (chunk, f)
        m.close()


def test_two_matchers_in_two_threads(vm, synth, monkeypatch):
    """instances are independent (the reference's Triangle globals are gone): two look-ahead runs at the same time, each
    with its own handle, host pool and streams, both with the final stage on the GPU share"""
    import threading
    import torch
    opt = {"dc_gpu": 1}
    monkeypatch.setenv("VSM_SEQ_CHUNK", "12")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 36
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    out = [None, None]

    def work(k):
        m = vm.Matcher(options=opt)
        m.set_intrinsics(*[float(x) for x in g["intr"]])
        for _ in range(3):
            out[k] = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
        m.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        for f in range(nf):
            assert len(out[k][f]) == int(g["counts"][f]) and G.sha(out[k][f]) == str(g["hashes"][f]), (k, f)


def test_sequence_api_fallbacks(vm, B, synth, monkeypatch):
    """what the look-ahead call takes frame by frame inside: mono input asked for stereo / quad matching (the reference's
    matchFeatures returns early on every frame), and sub-pixel refinement when the host-shared form is asked for"""
    seq = synth.stereo_sequence(8, 320, 128, 4)
    left = np.stack([l for l, _ in seq])
    right = np.stack([r for _, r in seq])
    for kw, meth, rgt, env in ((dict(refinement=2), 2, right, {"VSM_SEQ_V2": "0"}), (dict(), 2, None, {}), (dict(refinement=2), 1, right, {"VSM_SEQ_V2": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = vm.Matcher(**kw)
        got = g.run_sequence(left, rgt, meth)
        assert g.sequence_path() == 1
        c = B.CpuMatcher("oracle", **kw)
        for f, (l, r) in enumerate(seq):
            c.push_back(l, r if rgt is not None else None)
            c.match(meth)
            assert _same(got[f], c.matches()), (kw, f)
        g.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("method,mono", [(2, False), (1, False), (0, False), (0, True)])
def test_subpixel_refinement_in_the_batched_form(vm, B, synth, monkeypatch, method, mono):
    """refinement = 2 (parabolicFitting, viso/matcher.cpp:1379-1454: 7 x 7 costs, a 6 x 6 least-squares fit in double per
    match and step, matches whose fit fails dropped, :1541-1581) through the GPU-resident look-ahead form - fits, removal
    and the closing of the lists on the device, chunks of three so that chunk borders are crossed - against the oracle
    frame by frame, and against the per-frame path of the library"""
    monkeypatch.setenv("VSM_SEQ_CHUNK", "3")
    seq = synth.stereo_sequence(21, 480, 160, 8)
    left = np.stack([l for l, _ in seq])
    right = None if mono else np.stack([r for _, r in seq])
    g = vm.Matcher(refinement=2)
    got = g.run_sequence(left, right, method)
    assert g.sequence_path() == 2
    c = B.CpuMatcher("oracle", refinement=2)
    p = vm.Matcher(refinement=2)
    dropped = 0
    for f, (l, r) in enumerate(seq):
        c.push_back(l, None if mono else r)
        c.match(method)
        p.push_back(l, None if mono else r)
        p.match_features(method)
        assert _same(got[f], c.matches()), f
        assert _same(got[f], p.get_matches()), f
        dropped += 1 if f and len(got[f]) else 0
    assert dropped > 3   # (lists that are not empty: the test means something)
    g.close()
    p.close()


def test_degenerate_inputs(vm, B, synth):
    """no features / no cells / tiny images: same silent behaviour as the oracle, no faults"""
    flat = np.full((120, 200), 77, dtype=np.uint8)
    l, r = synth.stereo_sequence(2, 200, 120, 1)[0]
    for imgs in ([(flat, flat), (flat, flat)],                      # constant: no feature at all
                 [(l, r), (flat, flat), (l, r)],                    # features appear / vanish / reappear
                 [(l[:30, :40].copy(), r[:30, :40].copy())] * 2,    # too small for a single sparse cell
                 [(l[:14, :14].copy(), r[:14, :14].copy())] * 2):   # smaller than the margins
        g, c = vm.Matcher(stage_capture=True), B.CpuMatcher("oracle")
        for a, b in imgs:
            assert g.push_back(a, b) == 0
            c.push_back(a, b)
            for s in ("1c1", "1c2", "2c1", "2c2"):
                assert _same(g.features(s), c.features(s))
            for meth in (2, 1, 0):
                assert g.match(meth) == c.match(meth)
                assert _same(g.matches(), c.matches())
        got = vm.Matcher().run_sequence(np.stack([a for a, _ in imgs]), np.stack([b for _, b in imgs]), 2)
        c2 = B.CpuMatcher("oracle")
        for f, (a, b) in enumerate(imgs):
            c2.push_back(a, b)
            c2.match(2)
            assert _same(got[f], c2.matches())
        g.close()


def test_mono_then_stereo_and_replace_first(vm, B, synth):
    """ring-buffer corner cases: replace on an empty matcher, mono pushes between stereo pushes"""
    seq = synth.stereo_sequence(4, 320, 128, 4)
    g, c = vm.Matcher(stage_capture=True), B.CpuMatcher("oracle")
    steps = [(0, True, True), (1, True, False), (2, False, False), (3, True, False)]  # (frame, stereo?, replace?)
    for f, stereo, rep in steps:
        l, r = seq[f]
        g.push_back(l, r if stereo else None, replace=rep)
        c.push_back(l, r if stereo else None, replace=rep)
        for meth in (0, 1, 2):
            assert g.match(meth) == c.match(meth), (f, meth)
            assert _same(g.matches(), c.matches()), (f, meth)
    g.close()


# ---- stereo egomotion on top of the matcher ------------------------

def _load_golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


def test_vo_stereo_cases_golden(vm):
    G.replay_ego_cases(_load_golden("ego_cases"), vm.VisualOdometryStereo, vm.vo_sampler_seed)


@pytest.mark.parametrize("name", ["small_seq24_ego", "cfg2_seq200_ego"])
def test_vo_stereo_live_feedback_golden(vm, synth, name):
    """VisualOdometryStereo::process with LIVE Tr_delta feedback: flags, Tr_delta, bucketed lists and
    inlier sets of every frame equal the all-reference run's"""
    G.replay_ego_sequence(_load_golden(name), synth, vm.VisualOdometryStereo, vm.vo_sampler_seed)


def test_vo_stereo_vs_oracle_params_and_device_inputs(vm, B, synth):
    import torch
    w, h, n = 480, 200, 8
    seq = synth.stereo_sequence(31, w, h, n, disparity=14)
    intr = (420.0, w / 2.0, h / 2.0, 0.45)
    kw = dict(bucket=(3, 40.0, 30.0), ransac_iters=60, inlier_threshold=1.5, reweighting=False, nms_n=2, refinement=2)
    B.oracle_sampler_seed(71)
    o = B.OracleStereoVO(*intr, **kw)
    ref = []
    for f, (l, r) in enumerate(seq):
        ok, _, _, T = o.process(l, r, replace=(f == 5))
        ref.append((ok, T.copy(), o.bucketed().copy(), o.inliers().copy()))
    o.close()
    vm.vo_sampler_seed(71)
    v = vm.VisualOdometryStereo(*intr, **kw)
    for f, (l, r) in enumerate(seq):
        dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
        ok, _, _, T = v.process(dl, dr, replace=(f == 5))
        assert ok == ref[f][0], f
        assert T.tobytes() == ref[f][1].tobytes(), f
        assert _same(v.get_matches(), ref[f][2]), f
        assert np.array_equal(v.get_inlier_indices(), ref[f][3]), f
        assert v.get_number_of_matches() == len(ref[f][2]) and v.get_number_of_inliers() == len(ref[f][3])
    inl = v.get_inlier_indices()
    assert np.isfinite(v.get_gain(inl))
    v.close()


@pytest.mark.parametrize("K,device_inputs", [(8, True), (32, True), (3, False)])
def test_multi_sequence_lockstep_live_feedback(vm, synth, K, device_inputs):
    """vsm_multi_*: K independent stereo sequences stepped together with LIVE Tr_delta feedback (no recorded trail).  The
    sequences are config 4's eight seeds (repeated for K = 32); the fixture is the reference's VisualOdometryStereo on each
    of them in a process of its own (tests/golden/make_golden.py multi).  Per sequence and frame: what enters the matching
    (Tr_valid, Tr_delta), matchFeatures' final list (count + sha), process()'s flag, the bucketed list, the inlier set and
    the new Tr_delta."""
    import torch
    g = G.load("cfg4_multi_8procs_32f")
    w, h, nf = int(g["w"]), int(g["h"]), int(g["n_frames"])
    seeds = [int(s) for s in g["seeds"]][:min(K, 8)]
    canv = {sd: synth.canvas(sd, w, h) for sd in seeds}
    seq_seed = [seeds[k % len(seeds)] for k in range(K)]
    vo = vm.MultiVisualOdometryStereo(K, *[float(x) for x in g["intr"]])
    for f in range(nf):
        fr = {sd: synth.stereo_frame(canv[sd], f, w, h) for sd in seeds}
        left = np.stack([fr[sd][0] for sd in seq_seed])
        right = np.stack([fr[sd][1] for sd in seq_seed])
        tin = [(vo.motion_valid(k), vo.get_motion(k)) for k in range(K)]
        if device_inputs:
            ok = vo.process(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda())
        else:
            ok = vo.process(left, right)
        for k, sd in enumerate(seq_seed):
            key = f"s{sd}_"
            assert tin[k][0] == bool(g[key + "tr_valid"][f]), (f, k)
            if tin[k][0]:
                assert tin[k][1].tobytes() == g[key + "tr_in"][f].tobytes(), (f, k)
            fin, b, i = vo.get_matches(k, bucketed=False), vo.get_matches(k), vo.get_inlier_indices(k)
            assert len(fin) == int(g[key + "counts"][f]) and G.sha(fin) == str(g[key + "hashes"][f]), (f, k, len(fin))
            assert bool(ok[k]) == bool(g[key + "ok"][f]), (f, k)
            assert len(b) == int(g[key + "n_bucketed"][f]) and G.sha(b) == str(g[key + "bucketed_sha"][f]), (f, k)
            assert len(i) == int(g[key + "n_inliers"][f]) and G.sha(i) == str(g[key + "inliers_sha"][f]), (f, k)
            assert vo.get_motion(k).tobytes() == g[key + "tr_out"][f].tobytes(), (f, k)
    vo.close()


def test_multi_sequence_with_subpixel_refinement(vm, B, synth):
    """refinement = 2 inside vsm_multi_process, against the ORACLE's VisualOdometryStereo(refinement = 2) run on each sequence's
    images (a fresh object and sampler per sequence, like a process of its own): process()'s flag, Tr_delta, the bucketed
    list and the inlier set of every frame.  Besides: matchFeatures' un-bucketed final list equals what the per-frame matcher
    gives for the same images and the Tr_delta the step actually used; two sequences fed the same images stay identical to
    each other and to a run with K = 1"""
    w, h, nf, K = 640, 192, 6, 3
    canv = [synth.canvas(77, w, h), synth.canvas(78, w, h)]
    which = [0, 1, 0]
    intr = (645.24, 635.96, 194.13, 0.5707)
    want = []
    for cv in canv:
        B.oracle_sampler_seed(71)
        o = B.OracleStereoVO(*intr, refinement=2)
        rows = []
        for f in range(nf):
            l, r = synth.stereo_frame(cv, f, w, h)
            ok, _, _, T = o.process(l, r)
            rows.append((ok, T.copy(), o.bucketed().copy(), o.inliers().copy()))
        o.close()
        want.append(rows)
    vo = vm.MultiVisualOdometryStereo(K, *intr, refinement=2)
    solo = vm.MultiVisualOdometryStereo(1, *intr, refinement=2)
    per = [vm.Matcher(refinement=2) for _ in range(K)]
    for m in per:
        m.set_intrinsics(*intr)
    n_matched = 0
    for f in range(nf):
        fr = [synth.stereo_frame(cv, f, w, h) for cv in canv]
        left = np.stack([fr[i][0] for i in which])
        right = np.stack([fr[i][1] for i in which])
        tin = [(vo.motion_valid(k), vo.get_motion(k)) for k in range(K)]
        ok = vo.process(left, right)
        solo.process(left[:1], right[:1])
        for k in range(K):
            wk = want[which[k]][f]
            assert bool(ok[k]) == wk[0] and vo.get_motion(k).tobytes() == wk[1].tobytes(), (f, k)
            assert _same(vo.get_matches(k), wk[2]) and np.array_equal(vo.get_inlier_indices(k), wk[3]), (f, k)
            per[k].push_back(left[k], right[k])
            per[k].match_features(2, tin[k][1] if tin[k][0] else None)
            fin = vo.get_matches(k, bucketed=False)
            assert _same(fin, per[k].get_matches()), (f, k)
            n_matched += len(fin)
        assert _same(vo.get_matches(0, bucketed=False), vo.get_matches(2, bucketed=False)), f
        assert vo.get_motion(0).tobytes() == vo.get_motion(2).tobytes() == solo.get_motion(0).tobytes(), f
        assert _same(vo.get_matches(0), solo.get_matches(0)), f
    assert n_matched > 1000
    vo.close()
    solo.close()
    for m in per:
        m.close()


def test_multi_sequence_blank_frame_follows_the_reference(vm, B, synth):
    """A frame without features makes matchFeatures return early (viso/matcher.cpp:190-216): p_matched_2 keeps what the
    previous step's bucketFeatures left in it, process() buckets THAT list again (drawing from rand()) and hands it to
    updateMotion.  One of two lock-step sequences gets such a frame; both must follow the oracle's VisualOdometryStereo (run
    once per sequence, with the generators of a fresh process) through it and afterwards."""
    w, h, nf, K = 480, 200, 9, 2
    intr = (420.0, w / 2.0, h / 2.0, 0.45)
    seqs = [synth.stereo_sequence(41, w, h, nf, disparity=14), synth.stereo_sequence(42, w, h, nf, disparity=14)]
    blank = np.full((h, w), 90, dtype=np.uint8)
    seqs[0][4] = (blank, blank)
    seqs[0][5] = (blank, blank)   # two in a row: the list that is bucketed again has already been bucketed twice
    ref = []
    for k in range(K):
        B.oracle_sampler_seed(71)   # the state of viso/viso.cpp:88's function-local generator in a process of its own
        o = B.OracleStereoVO(*intr)
        rows = []
        for l, r in seqs[k]:
            ok, _, _, T = o.process(l, r)
            rows.append((ok, T.copy(), o.bucketed().copy(), o.inliers().copy()))
        ref.append(rows)
        o.close()
    assert len(ref[0][3][2]) > 20 and _same(ref[0][4][2], ref[0][4][2])
    vo = vm.MultiVisualOdometryStereo(K, *intr)
    for f in range(nf):
        ok = vo.process(np.stack([seqs[k][f][0] for k in range(K)]), np.stack([seqs[k][f][1] for k in range(K)]))
        for k in range(K):
            assert bool(ok[k]) == ref[k][f][0], (f, k)
            assert vo.get_motion(k).tobytes() == ref[k][f][1].tobytes(), (f, k)
            assert _same(vo.get_matches(k), ref[k][f][2]), (f, k, len(vo.get_matches(k)), len(ref[k][f][2]))
            assert np.array_equal(vo.get_inlier_indices(k), ref[k][f][3]), (f, k)
    vo.close()


def test_multi_sequence_failed_step_changes_nothing(vm, synth):
    """vsm_multi_process promises that a step which fails has advanced no sequence.  The late failure there is: sub-pixel
    refinement with more queries than its batched tail takes (VSM_PARA_MAX_LIST = 16384 dense features in the previous
    frame) - found only after the frame's features have gone through the kernels.  The failed call must leave motion, lists
    and the previous frame as they were, fail the same way when repeated, and a twin that is fed the frames again after a
    change of size (which restarts every sequence) must agree with a fresh object."""
    w, h = 1344, 720
    cv = synth.canvas(5, w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(3)]
    intr = (700.0, w / 2.0, h / 2.0, 0.5)
    vo = vm.MultiVisualOdometryStereo(1, *intr, refinement=2)
    one = lambda im: im[None]
    vo.process(one(fr[0][0]), one(fr[0][1]))
    state = lambda: (vo.motion_valid(0), vo.get_motion(0).tobytes(), vo.get_matches(0).tobytes(), vo.get_matches(0, bucketed=False).tobytes(),
                     vo.get_inlier_indices(0).tobytes())
    before = state()
    for _ in range(2):
        with pytest.raises(vm.VisoMatchError):
            vo.process(one(fr[1][0]), one(fr[1][1]))
        assert state() == before
    # a smaller image restarts the sequences; from there on the object behaves like a new one
    ws, hs = 480, 200
    seq = synth.stereo_sequence(43, ws, hs, 4, disparity=14)
    fresh = vm.MultiVisualOdometryStereo(1, *intr, refinement=2)
    for l, r in seq:
        a, b = vo.process(one(l), one(r)), fresh.process(one(l), one(r))
        assert bool(a[0]) == bool(b[0])
        assert vo.get_motion(0).tobytes() == fresh.get_motion(0).tobytes()
        assert _same(vo.get_matches(0), fresh.get_matches(0)) and _same(vo.get_matches(0, bucketed=False), fresh.get_matches(0, bucketed=False))
    assert len(vo.get_matches(0, bucketed=False)) > 100
    vo.close()
    fresh.close()


def test_multi_sequence_rejects_wrong_shapes_before_the_call(vm):
    vo = vm.MultiVisualOdometryStereo(3, 400.0, 100.0, 50.0, 0.5)
    im = np.zeros((2, 64, 96), dtype=np.uint8)   # K' = 2 < K = 3: the library would read a third image past the end
    with pytest.raises(vm.VisoMatchError):
        vo.process(im, im)
    with pytest.raises(vm.VisoMatchError):
        vo.process(np.zeros((3, 64, 96), dtype=np.uint8), np.zeros((3, 64, 80), dtype=np.uint8))
    vo.close()


# ---- monocular egomotion: HIP inlier counting + plane vote ----------

def test_vo_mono_cases_golden(vm):
    G.replay_mono_cases(_load_golden("mono_cases"), vm.VisualOdometryMono, vm.vo_sampler_seed)


def test_vo_mono_sequence_golden(vm, synth):
    G.replay_mono_sequence(_load_golden("mono_seq12_640x480"), synth, vm.VisualOdometryMono, vm.vo_sampler_seed)


def test_vo_mono_large_vs_oracle(vm, B):
    """enough matches for the GPU plane vote (>= 512 points in front of the camera) and 2000 hypotheses
    in one inlier-count launch; the host-only entry point must agree as well"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as MG
    rs = np.random.RandomState(99)
    f, cu, cv = MG.KITTI["f"], MG.KITTI["cu"], MG.KITTI["cv"]
    kw = dict(height=1.65, pitch=-0.08, ransac_iters=600)
    for n, motion in ((2500, (0.002, 0.012, -0.001, 0.03, -0.01, -0.9)), (700, (0.0, -0.02, 0.001, -0.05, 0.0, -0.6))):
        m = MG.mono_scene(rs, n, motion)
        B.oracle_sampler_seed(71)
        o = B.OracleMonoVO(f, cu, cv, **kw)
        ok_o, T_o = o.process_matches(m)
        inl_o = o.inliers()
        o.close()
        vm.vo_sampler_seed(71)
        v = vm.VisualOdometryMono(f, cu, cv, **kw)
        ok_v, T_v = v.process_matches(m)
        assert ok_v == ok_o and ok_o
        assert np.array_equal(v.get_inlier_indices(), inl_o)
        assert T_v.tobytes() == T_o.tobytes()
        v.close()
        vm.vo_sampler_seed(71)
        rc, _, T_h, inl_h = vm.host_estimate_motion_mono(m, vm.vo_mono_params(f, cu, cv, **kw), threads=4)
        assert rc == 1 and np.array_equal(inl_h, inl_o) and T_h.tobytes() == T_o.tobytes()


def test_vo_mono_dropin_header(vm, tmp_path):
    """tools/dropin/_build/mono_native = a C++ caller of include/viso_mono.h (VisualOdometryMono over
    the C-ABI): same result as the Python mirror on the same match list, both from a fresh sampler"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "dropin", "_build", "mono_native")
    if not os.path.exists(exe):
        pytest.skip("drop-in binary not built (needs the reference's matrix.cpp at build time)")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as MG
    m = MG.mono_scene(np.random.RandomState(3), 900, (0.001, 0.02, 0.0, 0.02, 0.0, -0.8))
    f, cu, cv = MG.KITTI["f"], MG.KITTI["cu"], MG.KITTI["cv"]
    src, out = tmp_path / "m.bin", tmp_path / "o.bin"
    with open(src, "wb") as fh:
        fh.write(np.array([len(m)], dtype=np.int32).tobytes())
        fh.write(m.tobytes())
    subprocess.check_call([exe, str(src), str(out), repr(f), repr(cu), repr(cv), "1.65", "-0.08", "300"], timeout=120)
    rec = np.fromfile(out, dtype=np.float64)
    vm.vo_sampler_seed(71)
    v = vm.VisualOdometryMono(f, cu, cv, height=1.65, pitch=-0.08, ransac_iters=300)
    ok, T = v.process_matches(m)
    assert bool(rec[0]) == ok and int(rec[1]) == v.get_number_of_inliers()
    assert rec[2:].reshape(4, 4).tobytes() == T.tobytes()
    v.close()


# ---- kernel-variant coverage: every NMS kernel, odd bin sizes for the fine-bin search, unaligned rows ---

@pytest.mark.parametrize("params", [
    dict(nms_n=1), dict(nms_n=2), dict(nms_n=3, multi_stage=0), dict(nms_n=4), dict(nms_n=7), dict(nms_n=11),
    dict(match_binsize=7), dict(match_binsize=23, match_radius=60), dict(match_binsize=100), dict(match_binsize=301),
    dict(half_resolution=0, nms_n=3), dict(half_resolution=0, nms_n=9, multi_stage=0, match_binsize=33),
    dict(match_radius=20, match_disp_tolerance=0), dict(nms_tau=10, outlier_flow_tolerance=2, outlier_disp_tolerance=2),
])
def test_kernel_variants_vs_oracle(vm, B, synth, params):
    """nms_n 3 / 9 (sparse of 3) take k_nms_fixed, 1,2,4 k_nms_tile, 5..10 k_nms_tile8, larger k_nms; the
    bin sizes exercise the fine-bin decomposition (VSM_VSUB sub-rows need not divide the bin)"""
    w, h = (333, 141) if params.get("half_resolution", 1) else (215, 117)
    seq = synth.stereo_sequence(91, w, h, 3, disparity=9, ramp=(1, 9))
    for method in (2, 0, 1):
        g, c = vm.Matcher(stage_capture=True, **params), B.CpuMatcher("oracle", **params)
        for f, (l, r) in enumerate(seq):
            g.push_back(l, r if method else None)
            c.push_back(l, r if method else None)
            for s in ("1c1", "1c2"):
                assert _same(g.features(s), c.features(s)), (params, method, f, s)
            assert g.match(method) == c.match(method)
            for s in range(5):
                assert _same(g.stage(s), c.stage(s)), (params, method, f, s)
        g.close()
        c.close()


def test_device_inputs_unaligned_rows_and_strides(vm, B, synth):
    """k_ingest: device images whose rows start at every byte alignment (width 1..3 mod 4, a strided
    view, and a view that starts in the middle of an allocation)"""
    import torch
    for w, h in ((201, 75), (202, 76), (203, 77), (204, 78)):
        seq = synth.stereo_sequence(13, w, h, 2, disparity=8)
        g, c = vm.Matcher(), B.CpuMatcher("oracle")
        for f, (l, r) in enumerate(seq):
            big = torch.zeros((2, h, w + 7), dtype=torch.uint8, device="cuda")   # row stride w+7, odd offsets
            big[0, :, 3:3 + w] = torch.from_numpy(l).cuda()
            big[1, :, 3:3 + w] = torch.from_numpy(r).cuda()
            assert g.push_back(big[0, :, 3:3 + w], big[1, :, 3:3 + w]) == 0
            c.push_back(l, r)
            for s in ("1c1", "1c2", "2c1", "2c2"):
                assert _same(g.features(s), c.features(s)), (w, f, s)
            assert g.match(2) == c.match(2)
            assert _same(g.get_matches(), c.matches()), (w, f)
        g.close()
        c.close()


def test_road_scene_golden(vm, synth):
    """street scene with real depth structure: the live stereo VO loop and the mono VO loop (success on
    every frame pair) against the all-reference run, frame by frame"""
    G.replay_road(_load_golden("road_1242x375"), synth, vm.VisualOdometryStereo, vm.VisualOdometryMono, vm.vo_sampler_seed)


@pytest.mark.parametrize("seed,heads", [(20240607, 0), (7, 0), (424242, 0), (7, 1)])
def test_randomised_configurations_vs_oracle(vm, B, synth, seed, heads):
    """seeded sweep over image sizes, parameters, methods and scene generators: every stage of the
    HIP path equals the oracle's (catches tile-border, bin-border and tie-break mistakes that fixed
    cases can miss); heads = 1: the second pass on per-bin head records (option match_heads)"""
    rs = np.random.RandomState(seed)
    pyr = synth.road_pyramid(77, levels=8, size=1024)
    for case in range(40):
        w = int(rs.randint(70, 700))
        h = int(rs.randint(60, 300))
        params = dict(nms_n=int(rs.choice([1, 2, 3, 3, 4, 5])), nms_tau=int(rs.choice([20, 50, 90])),
                      match_binsize=int(rs.choice([17, 32, 50, 64])), match_radius=int(rs.choice([40, 100, 200])),
                      match_disp_tolerance=int(rs.choice([1, 2, 3])), multi_stage=int(rs.choice([0, 1, 1])),
                      half_resolution=int(rs.choice([0, 1, 1])), refinement=int(rs.choice([0, 1, 1, 2])),
                      outlier_flow_tolerance=int(rs.choice([3, 5])), outlier_disp_tolerance=int(rs.choice([3, 5])))
        method = int(rs.choice([0, 1, 2, 2]))
        if rs.randint(2):
            seq = synth.stereo_sequence(int(rs.randint(1, 10000)), w, h, 3, disparity=int(rs.randint(2, 25)),
                                        ramp=(int(rs.randint(0, 3)), int(rs.randint(8, 30))))
        else:
            seq = [synth.road_stereo_frame(pyr, f, w, h, step_mm=int(rs.choice([150, 400])), base_mm=300) for f in range(3)]
        use_tr = method == 2 and rs.randint(2) == 1
        Tr = np.eye(4)
        Tr[2, 3] = -0.3
        g, c = vm.Matcher(stage_capture=True, options={"match_heads": heads}, **params), B.CpuMatcher("oracle", **params)
        intr = (400.0, w / 2.0, h / 2.0, 0.3)
        g.set_intrinsics(*intr)
        c.set_intrinsics(*intr)
        for f, (l, r) in enumerate(seq):
            rep = f == 2 and rs.randint(3) == 0
            g.push_back(l, r if method else None, replace=rep)
            c.push_back(l, r if method else None, replace=rep)
            for s in ("1c1", "1c2"):
                assert _same(g.features(s), c.features(s)), (case, params, f, s)
            t = Tr if (use_tr and f >= 1) else None
            assert g.match(method, t) == c.match(method, t), (case, params, f)
            for s in range(5):
                assert _same(g.stage(s), c.stage(s)), (case, w, h, params, method, f, s)
        g.close()
        c.close()


def test_emulated_vertex_sort_on_gpu(vm):
    """k_dc_ties (Triangle's randomised vertex sort on one wave, in LDS) names the same match for every shared pixel
    as the host emulation: sizes around the 64-lane boundary, heavy duplication, all points equal"""
    rs = np.random.RandomState(7)
    cases = []
    for n in (2, 3, 5, 8, 33, 63, 64, 65, 66, 129, 500, 2048, 7400, 8192):
        for span in (2, 30, 400):
            cases.append(np.stack([rs.randint(0, span * 2 + 1, n) * 2, rs.randint(0, span + 1, n) * 2], 1))
    g = np.stack(np.meshgrid(np.arange(0, 60, 2), np.arange(0, 40, 2)), -1).reshape(-1, 2)
    cases += [np.concatenate([g, g[::3]]), np.zeros((70, 2), dtype=np.int64)]
    checked = 0
    for p in cases:
        host, _ = vm.ties(p, gpu=False)
        dev, _ = vm.ties(p, gpu=True)
        if dev is None:          # more shared pixels than the kernel reports (255): it declines, the host decides
            assert len(host) > 255, len(p)
            continue
        assert np.array_equal(host, dev), len(p)
        checked += 1
    assert checked > 30


def test_delaunay_subtrees_on_gpu(vm, B):
    """the shared form of the exact Delaunay: sub-trees triangulated by GPU threads (vsm_dc.hip), the
    host continues on the same arrays; triangle sets equal the ORACLE's Triangle restatement (viso/triangle.cpp:6161
    as the oracle restates it) for every split"""
    import time
    rs = np.random.RandomState(3)

    def canon(t):
        t = np.sort(np.asarray(t), axis=1)
        return t[np.lexsort(t.T[::-1])]

    cases = [np.stack([rs.randint(0, 620, n) * 2, rs.randint(0, 187, n) * 2], 1) for n in (5, 64, 700, 7400)]
    g = np.stack(np.meshgrid(np.arange(0, 60, 2), np.arange(0, 40, 2)), -1).reshape(-1, 2)
    cases += [g, np.concatenate([g, g[::3]]), np.stack([np.arange(0, 300, 2), np.full(150, 8)], 1)]
    big = np.stack([rs.randint(0, 1024, 20000) * 2, rs.randint(0, 512, 20000) * 2], 1)   # > 16384 distinct points
    def oracle_tris(p):
        return canon(B.delaunay("oracle", p.astype(np.float32)))

    assert np.array_equal(oracle_tris(big), canon(vm.delaunay_gpu_split(big, 480, -1, True)))
    for p in cases:
        whole = oracle_tris(p)
        assert np.array_equal(whole, canon(vm.host_delaunay(p, 1))), len(p)
        for leaf, top in ((3, 0), (14, 0), (56, 0), (500, 0), (3, 12), (14, 120), (16, 240), (56, 480), (30, 900),
                          (480, -1), (100, -1), (14, -1), (5000, -1)):
            for kd in (False, True):
                assert np.array_equal(whole, canon(vm.delaunay_gpu_split(p, leaf, top, kd))), (len(p), leaf, top, kd)


def _oracle_survivors(B, lst, method):
    """Matcher::removeOutliers (viso/matcher.cpp:1207-1377) on a match list: the CPU oracle's (pinned against the reference's
    own on these very lists by the CPU suite, tests/test_oracle_vs_ref.py) and, where the compiled reference travelled to
    this box (oracle/_ref), the reference's as well - both must agree.
    One kind of list never goes to the reference: more than three matches that all share ONE pixel.  Its vendored Triangle
    then recurses without end (viso/triangle.cpp:5966-6092: divconqrecurse on a single distinct vertex halves it into 0 + 1
    and calls itself on the 1) until the stack is gone - the segmentation fault round 4 recorded in this test and put down
    to symbol clashes with the HIP runtime.  It happens in a process without any GPU library just the same
    (tests/test_oracle_vs_ref.py::test_reference_crashes_on_a_one_pixel_list); the oracle and this library define the case:
    no triangulation, no support, nothing survives."""
    want = B.remove_outliers("oracle", lst, method)
    distinct = len(set(zip(np.asarray(lst["u1c"]).tolist(), np.asarray(lst["v1c"]).tolist()))) if len(lst) else 0
    if B.have_ref() and not (len(lst) > 3 and distinct < 2):
        assert _same(want, B.remove_outliers("ref", lst, method))
    return want


def test_gpu_resident_remove_outliers_chain(vm, B):
    """the device chain of the GPU-resident look-ahead form (keys, (x,y) sort + duplicates + kd order, block sub-trees on the
    edge-word LDS mesh, merge levels through the band cache, tie patches from the device's or the host's vertex sort,
    support votes, survivors, prior statistics): the survivors against the ORACLE's removeOutliers, byte for byte; the prior boxes against the host code of the per-frame path (whose boxes
    the stage goldens pin).  List lengths around every structural boundary (3 | 4, five points - the case the vectorizers
    once broke -, one block | two, small | large bands)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dc2_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dc2_check.py"))
    dc2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc2)
    for n in (0, 3, 4, 5, 17, 100, 480, 481, 961, 2000, 4500, 7400, 9000):
        for method in (0, 1, 2):
            for grid in (False, True):
                lst = dc2.make_list(n, grid=grid)
                want = _oracle_survivors(B, lst, method)
                hs, hr, _ = vm.remove_outliers(lst, method, 1242, 375)
                assert _same(want, hs), (n, method, grid, len(want), len(hs))
                for gt in (False, True):
                    if gt and n > 8192:
                        continue
                    gs, gr, _ = vm.remove_outliers(lst, method, 1242, 375, gpu=True, gpu_ties=gt, copies=2)
                    assert _same(want, gs) and np.array_equal(hr, gr), (n, method, grid, gt, len(want), len(gs))
    lst = dc2.make_list(60, dup=0)
    lst["u1c"], lst["v1c"] = 100, 50                       # every match at one pixel: no triangulation, nothing survives
    assert len(vm.remove_outliers(lst, 2, 1242, 375, gpu=True)[0]) == len(_oracle_survivors(B, lst, 2)) == 0
    lst = dc2.make_list(300, dup=0)
    lst["v1c"] = 40                                          # collinear
    assert _same(vm.remove_outliers(lst, 2, 1242, 375, gpu=True)[0], _oracle_survivors(B, lst, 2))


def test_device_chain_merge_nodes_that_overflow_their_band(vm, B):
    """a merge node of the device chain caches the records near its cut (the band) in LDS; one that needs more lines than
    the launch gave it is redone by a single lane on the records in global memory.  No list of the benchmark gets there
    with the default sizing, so the sizing is turned down until the large nodes do (vsm_debug_dc2_band_factor(0): 256 lines
    whatever the node): survivors against the oracle's removeOutliers, lattice lists (every circumcircle through four
    points) included"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dc2_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dc2_check.py"))
    dc2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc2)
    try:
        for f in (0, 1, 3):
            vm.lib().vsm_debug_dc2_band_factor(f)
            for n in (2000, 7400, 9000):
                for grid in (False, True):
                    lst = dc2.make_list(n, grid=grid)
                    want = _oracle_survivors(B, lst, 2)
                    gs, _, _ = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, copies=2)
                    assert _same(want, gs), (f, n, grid, len(want), len(gs))
    finally:
        vm.lib().vsm_debug_dc2_band_factor(-1)


def test_device_chain_at_the_capacity_steps_of_its_lds_kernels(vm, B, monkeypatch):
    """the LDS preparation kernel (sort, duplicates, kd order) is launched with a power-of-two capacity and capacity / 4 or / 8
    threads, the chain's last kernel keeps its tables in LDS up to 12288 matches: list lengths on both sides of every step,
    heavy duplicates included - survivors against the oracle's removeOutliers, prior boxes against the host code.  Lists beyond
    8192 matches both ways: their own kernel (every thread at work), and the narrow form inside the LDS kernel that a handle
    uses until it has met such a list (VSM_DC2_EXPECT_LONG=0)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dc2_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dc2_check.py"))
    dc2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc2)
    for n in (255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4095, 4096, 4097, 8191, 8192, 8193, 12287, 12288, 12289):
        for grid in (False, True):
            lst = dc2.make_list(n, grid=grid)
            want = _oracle_survivors(B, lst, 2)
            hs, hr, _ = vm.remove_outliers(lst, 2, 1242, 375)
            assert _same(want, hs), (n, grid, len(want), len(hs))
            for gt in (False, True):
                if gt and n > 8192:
                    continue
                gs, gr, _ = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=gt, copies=3)
                assert _same(want, gs) and np.array_equal(hr, gr), (n, grid, gt, len(want), len(gs))
            if n > 8192:
                monkeypatch.setenv("VSM_DC2_EXPECT_LONG", "0")
                gs, gr, _ = vm.remove_outliers(lst, 2, 1242, 375, gpu=True, gpu_ties=False, copies=3)
                monkeypatch.delenv("VSM_DC2_EXPECT_LONG")
                assert _same(want, gs) and np.array_equal(hr, gr), (n, grid, "narrow", len(want), len(gs))


@pytest.mark.parametrize("form", ["host-shared", "GPU-resident"])
def test_all_eight_sequences_of_config_4_on_one_gpu(vm, synth, monkeypatch, form):
    """config 4 = eight independent sequences, one per GPU.  Without the 8-GPU node every rank's sequence still has to go
    through the HIP path somewhere: seeds 1234..1241, 40 frames each, look-ahead API, final lists against the
    reference's hashes (tests/golden/cfg4_seq200_tr_8seeds.npz)"""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "1" if form == "GPU-resident" else "0")
    opt = {"dc_gpu": 1}
    monkeypatch.setenv("VSM_SEQ_CHUNK", "20")
    g = G.load("cfg4_seq200_tr_8seeds")
    w, h, nf = 1242, 375, 40
    m = vm.Matcher(options=opt)
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    for seed in range(1234, 1242):
        key = f"s{seed}"
        cv = synth.canvas(seed, w, h)
        fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
        left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
        right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
        tr = np.ascontiguousarray(g[key + "_tr_in"][:nf].reshape(nf, 16)[:, :12])
        got = m.run_sequence(left, right, 2, tr, g[key + "_tr_valid"][:nf])
        for f in range(nf):
            assert len(got[f]) == int(g[key + "_counts"][f]) and G.sha(got[f]) == str(g[key + "_hashes"][f]), (seed, f)
    m.close()


def test_inputs_produced_asynchronously_on_another_stream(vm, B, synth):
    """device-resident inputs are read on the handle's own stream: the wrapper orders that stream behind torch's current
    one (vsm_wait_for_stream), so images that a side stream is still writing when push_back / run_sequence are called -
    behind a long-running kernel - arrive complete"""
    import torch
    seq = synth.stereo_sequence(5, 500, 200, 4)
    left = np.stack([l for l, _ in seq])
    right = np.stack([r for _, r in seq])
    c = B.CpuMatcher("oracle")
    want = []
    for l, r in seq:
        c.push_back(l, r)
        c.match(2)
        want.append(c.matches())
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    hl, hr = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
    g = vm.Matcher()
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big @ big * 1e-4                       # tens of milliseconds of work in front of the copies
        dl, dr = hl.to("cuda", non_blocking=True), hr.to("cuda", non_blocking=True)
        got = g.run_sequence(dl, dr, 2)                  # called while the copies are still queued
    for f in range(len(seq)):
        assert _same(got[f], want[f]), f
    g2 = vm.Matcher()
    with torch.cuda.stream(side):
        for f in range(len(seq)):
            for _ in range(5):
                big = big @ big * 1e-4
            a, b = hl[f].to("cuda", non_blocking=True), hr[f].to("cuda", non_blocking=True)
            assert g2.push_back(a, b) == 0
            g2.match(2)
            assert _same(g2.matches(), want[f]), f
    torch.cuda.synchronize()
    g.close()
    g2.close()


def test_lost_completion_callback_of_the_delaunay_share(vm, synth, monkeypatch):
    """host-shared look-ahead form, fault injection: the host function that reports the end of the GPU's share of a
    chunk's Delaunay stage never runs (option dc_fault_inject = 1).  dc_wait()'s watchdog (shortened to 0.3 s) synchronises the
    stream, finds it healthy and carries on with the device's results: same lists, no error"""
    import torch
    monkeypatch.setenv("VSM_SEQ_V2", "0")
    opt = {"dc_gpu": 1, "dc_fault_inject": 1, "dc_watchdog_ms": 300}
    monkeypatch.setenv("VSM_SEQ_CHUNK", "10")
    g = G.load("cfg2_seq200_tr")
    w, h, nf = int(g["w"]), int(g["h"]), 20
    cv = synth.canvas(int(g["seed"]), w, h)
    fr = [synth.stereo_frame(cv, f, w, h) for f in range(nf)]
    left = torch.from_numpy(np.stack([l for l, _ in fr])).cuda()
    right = torch.from_numpy(np.stack([r for _, r in fr])).cuda()
    m = vm.Matcher(options=opt)
    m.set_intrinsics(*[float(x) for x in g["intr"]])
    got = m.run_sequence(left, right, 2, g["tr_in"][:nf], g["tr_valid"][:nf])
    for f in range(nf):
        assert len(got[f]) == int(g["counts"][f]) and G.sha(got[f]) == str(g["hashes"][f]), f
    m.close()
