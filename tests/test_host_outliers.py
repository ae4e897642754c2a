"""CPU suite: the host code of Matcher::removeOutliers (viso/matcher.cpp:1207-1377) as the per-frame path runs it - one
triangulation split over fork-join threads, started from the packed pixels, flows / votes / survivors' copy split over the
threads (vsm_host_outliers_and_prior_threads, no GPU needed) - against the oracle's removeOutliers and, where its build is
present, the reference's; the prior boxes of the threaded and the plain form against each other."""
import numpy as np
import pytest

from conftest import pkg


def _list(vm, rs, n, w=1242, h=375, dup=0.02, grid=False):
    m = np.zeros(n, dtype=vm.P_MATCH)
    if grid:
        k = int(np.ceil(np.sqrt(max(n, 1))))
        g = np.stack(np.meshgrid(np.arange(20, 20 + 2 * k, 2), np.arange(20, 20 + 2 * k, 2)), -1).reshape(-1, 2)[:n]
        u, v = g[:, 0].copy(), g[:, 1].copy()
    else:
        u = rs.randint(6, w // 2 - 6, n) * 2
        v = rs.randint(6, h // 2 - 6, n) * 2
    nd = int(n * dup)
    if nd and n > 10:
        src, dst = rs.randint(0, n, nd), rs.randint(0, n, nd)
        u[dst], v[dst] = u[src], v[src]
    m["u1c"], m["v1c"] = u, v
    fl = rs.randint(-3, 4, (n, 2))
    bad = rs.rand(n) < 0.1
    fl[bad] += rs.randint(-30, 30, (int(bad.sum()), 2))
    m["u1p"], m["v1p"] = u + 6 + fl[:, 0], v + fl[:, 1]
    d = 20 + rs.randint(-2, 3, n)
    d[rs.rand(n) < 0.05] += 17
    m["u2c"], m["v2c"] = u - d, v
    m["u2p"], m["v2p"] = m["u1p"] - d - rs.randint(-1, 2, n), m["v1p"]
    for k in ("i1p", "i2p", "i1c", "i2c"):
        m[k] = rs.randint(0, 9000, n)
    return m


def _same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("threads", [2, 8])
def test_threaded_remove_outliers_equals_oracle(B, threads):
    vm = pkg("visomatch")
    rs = np.random.RandomState(11)
    # (2048 is where flows, votes and the survivors' copy go to the threads; 3 | 4 where the reference starts to triangulate)
    for n in (0, 3, 4, 5, 100, 870, 2047, 2048, 2049, 4500, 7400):
        for grid in (False, True):
            lst = _list(vm, rs, n, grid=grid)
            for method in (0, 1, 2):
                want = B.remove_outliers("oracle", lst, method)
                plain, rg1, _ = vm.remove_outliers(lst, method, 1242, 375)
                got, rgt, _ = vm.remove_outliers(lst, method, 1242, 375, threads=threads)
                assert _same(want, plain), (n, grid, method, len(want), len(plain))
                assert _same(want, got), (n, grid, method, threads, len(want), len(got))
                assert np.array_equal(rg1, rgt), (n, grid, method)


def test_threaded_remove_outliers_equals_reference(B):
    if not B.have_ref():
        pytest.skip("oracle/_ref not built")
    vm = pkg("visomatch")
    rs = np.random.RandomState(12)
    for n in (870, 3000, 7400):
        lst = _list(vm, rs, n)
        assert _same(B.remove_outliers("ref", lst, 2), vm.remove_outliers(lst, 2, 1242, 375, threads=8)[0]), n
