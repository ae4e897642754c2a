"""CPU suite: the product's host Delaunay (libvisomatch.so, no GPU needed) against the oracle's and,
where its build is present, the real Triangle of the reference -- sequential and multi-threaded."""
import numpy as np
import pytest

from conftest import pkg


def _tri_set(t):
    return sorted(tuple(sorted(map(int, r))) for r in t)


def _cases(seed):
    rng = np.random.default_rng(seed)
    out = []
    for n, span in ((4, 3), (5, 4), (7, 3), (12, 4), (50, 8), (200, 12), (300, 200), (500, 40), (3000, 120), (8000, 600), (20000, 1000)):
        out.append(np.stack([rng.integers(0, span * 2, n) * 2, rng.integers(0, span, n) * 2], 1))
    g = np.stack(np.meshgrid(np.arange(0, 60, 2), np.arange(0, 50, 2)), -1).reshape(-1, 2)
    out += [g, g[rng.permutation(len(g))], np.stack([np.arange(40) * 2, np.full(40, 6)], 1), np.concatenate([g[:300], g[:300]])]
    out.append(np.stack([np.full(300, 8), np.arange(300) * 2], 1))  # vertical line
    return out


@pytest.mark.parametrize("threads", [1, 2, 4, 8])
@pytest.mark.parametrize("seed", range(3))
def test_product_delaunay_equals_oracle(B, seed, threads):
    vm = pkg("visomatch")
    for pts in _cases(seed):
        a = vm.host_delaunay(pts, threads=threads)
        b = B.delaunay("oracle", pts.astype(np.float32))
        assert len(a) == len(b)
        assert _tri_set(a) == _tri_set(b)


def test_product_delaunay_equals_reference(B):
    if not B.have_ref():
        pytest.skip("oracle/_ref not built")
    vm = pkg("visomatch")
    for pts in _cases(11):
        assert _tri_set(vm.host_delaunay(pts, threads=4)) == _tri_set(B.delaunay("ref", pts.astype(np.float32)))


def test_plain_mask_pass_of_the_sort(B):
    """the emulated vertex sort writes its two bit masks with AVX-512 where the CPU has it; the plain loop
    (VSM_NO_AVX512=1, read when the library is loaded, hence the child process) must make the same decisions"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from conftest import pkg\n"
        "import test_host_delaunay as T\n"
        "vm = pkg('visomatch')\n"
        "np.save(sys.argv[1], np.array([vm.host_delaunay(p, threads=1) for p in T._cases(5)], dtype=object), allow_pickle=True)\n"
    ) % os.path.dirname(os.path.abspath(__file__))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "t.npy")
        env = dict(os.environ, VSM_NO_AVX512="1")
        subprocess.check_call([sys.executable, "-c", code, out], env=env, timeout=300)
        plain = np.load(out, allow_pickle=True)
    vm = pkg("visomatch")
    for pts, t in zip(_cases(5), plain):
        assert _tri_set(vm.host_delaunay(pts, threads=1)) == _tri_set(t)
        assert _tri_set(t) == _tri_set(B.delaunay("oracle", pts.astype(np.float32)))


def test_split_form_equals_whole(B):
    """prepare / independent sub-trees / merges (the form shared between host and GPU in the look-ahead
    path) gives the same triangle set as the one-piece run for every sub-tree size, duplicates and
    degenerate layouts included"""
    vm = pkg("visomatch")
    rs = np.random.RandomState(11)

    def canon(t):
        t = np.sort(np.asarray(t), axis=1)
        return t[np.lexsort(t.T[::-1])]

    cases = []
    for n in (2, 3, 5, 17, 64, 300, 2500):
        p = np.stack([rs.randint(0, 120, n) * 2, rs.randint(0, 50, n) * 2], 1)
        cases.append(p)
    g = np.stack(np.meshgrid(np.arange(0, 40, 2), np.arange(0, 24, 2)), -1).reshape(-1, 2)   # lattice: all co-circular
    cases.append(g)
    cases.append(np.concatenate([g, g[::3]]))                                                    # + duplicates
    cases.append(np.stack([np.arange(0, 200, 2), np.full(100, 8)], 1))                            # collinear
    for p in cases:
        whole = vm.host_delaunay(p, 1)
        for leaf in (2, 3, 4, 7, 14, 56, 100000):
            for top in (0, 4 * leaf, 100000, -1):
                part = vm.host_delaunay_split(p, leaf, top)
                assert np.array_equal(canon(whole), canon(part)), (len(p), leaf, top)
