"""The product's monocular egomotion (csrc/vsm_mono.hip + vsm_linalg.h) through its host-only entry
point (no GPU: inlier counting and the plane vote run on host threads) against the golden vectors
recorded from the reference."""
import os

import numpy as np
import pytest

from conftest import pkg

HERE = os.path.dirname(os.path.abspath(__file__))


class HostMonoVO:
    """VisualOdometry::process(matches) semantics on top of the host-only solver entry point"""

    def __init__(self, vm, threads, f, cu, cv, **mono):
        self.vm, self.threads = vm, threads
        self.par = vm.vo_mono_params(f, cu, cv, **mono)
        self.T = np.eye(4)
        self.inl = np.zeros(0, dtype=np.int32)

    def process_matches(self, m):
        rc, _, T, inl = self.vm.host_estimate_motion_mono(m, self.par, self.threads)
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
def test_mono_cases_golden(threads):
    import golden_util as G
    vm = pkg("visomatch")
    g = np.load(os.path.join(HERE, "golden", "mono_cases.npz"))
    G.replay_mono_cases(g, lambda *a, **k: HostMonoVO(vm, threads, *a, **k), vm.vo_sampler_seed)
