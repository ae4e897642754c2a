"""CPU suite: the oracle (oracle/viso_oracle.c) must reproduce every committed golden vector
(tests/golden/*.npz = outputs of the real reference, see tests/golden/make_golden.py)."""
import pytest

import golden_util as G


def _oracle(B):
    return lambda **p: B.CpuMatcher("oracle", **p)


@pytest.mark.parametrize("method", [2, 0, 1])
@pytest.mark.parametrize("pi", range(6))
def test_small_full_dumps(B, synth, pi, method):
    g = G.load("small_304x128")
    assert G.replay_small(g, synth, _oracle(B), pi, method) > 5


@pytest.mark.parametrize("pi", range(2))
def test_small_tr_delta(B, synth, pi):
    g = G.load("small_tr_352x160")
    assert G.replay_small(g, synth, _oracle(B), pi, 2) > 5


def test_cfgA_hashes(B, synth):
    G.replay_hashed(G.load("cfgA_1242x375_quad"), synth, _oracle(B))


def test_cfg3_mono_hashes(B, synth):
    G.replay_hashed(G.load("cfg3_640x480_mono"), synth, _oracle(B), frames=3)


def test_cfg5_highres_hashes(B, synth):
    G.replay_hashed(G.load("cfg5_2048x1024_quad"), synth, _oracle(B))


def test_cfg5_as_specified_20k_dense_features_hashes(B, synth):
    """config 5 with the blur radius (29) at which a 2048x1024 image has ~20 k dense features, as BASELINE.json words it"""
    g = G.load("cfg5_2048x1024_quad_20k")
    assert 19000 < int(g["counts"][0][1]) < 21500
    G.replay_hashed(g, synth, _oracle(B), frames=3)


def test_cfg2_sequence_with_feedback(B, synth):
    G.replay_vo_sequence(G.load("cfg2_seq200_tr"), synth, _oracle(B), n_frames=6)
