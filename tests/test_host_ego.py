"""The product's egomotion solver (csrc/vsm_ego.cpp, host code, no GPU needed) against the golden
vectors recorded from the reference and against the oracle."""
import os

import numpy as np
import pytest

from conftest import pkg

HERE = os.path.dirname(os.path.abspath(__file__))


class HostEgoVO:
    """VisualOdometry::process(matches) semantics on top of the host-only solver entry point"""

    def __init__(self, vm, threads, f, cu, cv, base, **ego):
        self.vm, self.threads = vm, threads
        self.par = vm.vo_stereo_params(f, cu, cv, base, **ego)
        self.T = np.eye(4)
        self.inl = np.zeros(0, dtype=np.int32)

    def process_matches(self, m):
        rc, _, T, inl = self.vm.host_estimate_motion_stereo(m, self.par, self.threads)
        if inl is not None:
            self.inl = inl
        if rc == 1:
            self.T = T
        return rc == 1, self.T

    def inliers(self):
        return self.inl

    def close(self):
        pass


@pytest.mark.parametrize("threads", [1, 4])
def test_ego_cases_golden(threads):
    import golden_util as G
    vm = pkg("visomatch")
    g = np.load(os.path.join(HERE, "golden", "ego_cases.npz"))
    G.replay_ego_cases(g, lambda *a, **k: HostEgoVO(vm, threads, *a, **k), vm.vo_sampler_seed)


def test_ego_vs_oracle_random(B):
    vm = pkg("visomatch")
    rs = np.random.RandomState(9)
    f, cu, cv, base = 480.0, 300.0, 110.0, 0.3
    vm.vo_sampler_seed(71)
    B.oracle_sampler_seed(71)
    for case in range(40):
        n = int(rs.choice([6, 9, 30, 120, 500]))
        X, Y, Z = rs.uniform(-6, 6, n), rs.uniform(-2, 2, n), rs.uniform(2, 30, n)
        ry, tz = rs.uniform(-0.05, 0.05), rs.uniform(-1.0, 0.3)
        Xc, Zc = np.cos(ry) * X + np.sin(ry) * Z, -np.sin(ry) * X + np.cos(ry) * Z + tz
        m = np.zeros(n, dtype=B.MATCH_DTYPE)
        m["u1p"], m["v1p"], m["u2p"], m["v2p"] = f * X / Z + cu, f * Y / Z + cv, f * (X - base) / Z + cu, f * Y / Z + cv
        m["u1c"], m["v1c"] = f * Xc / Zc + cu, f * Y / Zc + cv
        m["u2c"], m["v2c"] = f * (Xc - base) / Zc + cu, f * Y / Zc + cv
        bad = rs.permutation(n)[: int(rs.choice([0, 0.3, 0.7]) * n)]
        for k in ("u1c", "v1c", "u2c", "v2c"):
            m[k] += rs.normal(0, 0.4, n).astype(np.float32)
            m[k][bad] += rs.uniform(-25, 25, len(bad)).astype(np.float32)
        ego = dict(ransac_iters=int(rs.choice([3, 40, 200])), inlier_threshold=float(rs.choice([1.0, 2.0])),
                   reweighting=bool(rs.randint(2)))
        rc_o, tr_o, inl_o = B.oracle_estimate_motion(m, B.ego_params(f, cu, cv, base, **ego))
        rc_p, tr_p, _, inl_p = vm.host_estimate_motion_stereo(m, vm.vo_stereo_params(f, cu, cv, base, **ego),
                                                              threads=1 + case % 3)
        assert rc_o == rc_p, case
        assert np.array_equal(inl_o, inl_p), case
        if rc_o == 1:
            assert tr_o.tobytes() == tr_p.tobytes(), case
