"""CPU emulation of the fused matching-resolution kernels (k_feat_dense / k_feat_sparse) against the oracle.

`tests/emu/feat_emu.cpp` walks the per-thread functions of `csrc/vsm_feat.h` - the very code the gfx950 kernels run -
tile by tile and thread by thread.  Compared with the oracle (itself pinned against the reference's filter.o /
matcher.o in test_oracle_vs_ref.py): whole du / dv / f1 / f2 planes, and the survivors of both suppression scales in the
reference's emission order (viso/matcher.cpp:344-430).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
CSRC = os.path.join(os.path.dirname(HERE), "opencl-structure-from-motion_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emu():
    so, src, hdr = (os.path.join(EMU, "libfeatemu.so"), os.path.join(EMU, "feat_emu.cpp"), os.path.join(CSRC, "vsm_feat.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + CSRC, src, "-o", so])
    return so


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with vector extensions here")
    L = C.CDLL(build_emu())
    vp = C.c_void_p
    L.emu_feat_dense.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.emu_feat_sparse.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    return L


def cells(length, n, margin=6):
    span = length - 2 * n - 2 * margin
    return (span + n) // (n + 1) if span > 0 else 0


def cand_list(cand, ncu, ncv):
    """survivors in the reference's emission order: cells u-major / v-minor, classes f1min, f1max, f2min, f2max"""
    out = []
    c = cand.reshape(ncu * ncv, 4)
    for cell in range(ncu * ncv):
        for g in range(4):
            v = int(c[cell, g])
            if v < 0:
                v &= 0xFFFFFFFF
                out.append((v & 0x3FFF, (v >> 14) & 0x3FFF, g))
    return out


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


CASES = [(1242, 375, True, 50), (640, 480, False, 50), (320, 96, True, 20), (304, 128, False, 50), (2048, 1024, True, 50),
         (352, 160, True, 50), (200, 70, False, 10), (1242, 375, False, 50)]


@pytest.mark.parametrize("w,h,half,tau", CASES)
def test_fused_tiles_vs_oracle(emu, B, synth, w, h, half, tau):
    l, _ = synth.stereo_sequence(33, w, h, 1)[0]
    img = B.pad_image(l)
    mimg = np.ascontiguousarray(B.half_image("oracle", img, w) if half else img)
    mw, mh = (w // 2, h // 2) if half else (w, h)
    mbpl = mimg.shape[1]
    assert mimg.shape[0] == mh
    du_o, dv_o = B.sobel5x5("oracle", mimg)
    f1_o, f2_o = B.blob5x5("oracle", mimg), B.checkerboard5x5("oracle", mimg)
    du, dv = np.full((mh, mbpl), 77, np.uint8), np.full((mh, mbpl), 77, np.uint8)
    f1, f2 = np.full((mh, mbpl), 777, np.int16), np.full((mh, mbpl), 777, np.int16)
    ncu, ncv = cells(mw, 3), cells(mh, 3)
    cand = np.full(max(ncu * ncv, 1) * 4, 12345, np.int32)
    emu.emu_feat_dense(ptr(mimg), mw, mh, mbpl, tau, ncu, ncv, ptr(du), ptr(dv), ptr(f1), ptr(f2), ptr(cand))
    assert np.array_equal(du, du_o) and np.array_equal(dv, dv_o)
    assert np.array_equal(f1, f1_o) and np.array_equal(f2, f2_o)
    want = [tuple(int(x) for x in (r[0], r[1], r[3])) for r in B.nms("oracle", f1_o, f2_o, mw, 3, tau)]
    assert cand_list(cand, ncu, ncv) == want and len(want) > 20
    # sparse scale
    ncu9, ncv9 = cells(mw, 9), cells(mh, 9)
    if ncu9 * ncv9 > 0:
        cand9 = np.full(ncu9 * ncv9 * 4, 12345, np.int32)
        emu.emu_feat_sparse(ptr(mimg), mw, mh, mbpl, tau, ncu9, ncv9, ptr(cand9))
        want9 = [tuple(int(x) for x in (r[0], r[1], r[3])) for r in B.nms("oracle", f1_o, f2_o, mw, 9, tau)]
        assert cand_list(cand9, ncu9, ncv9) == want9
