"""TEST INFRASTRUCTURE ONLY: ctypes bindings for

  * liboracle.so          -- the CPU restatement (oracle/viso_oracle.c)
  * _ref/libvisoref.so    -- the real reference compiled in place (oracle/Makefile `make ref`)

Both expose the same Python surface (class CpuMatcher) so that tests can run one
body against either.  Nothing in the product imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.environ.get("VISO_REF_SO") or os.path.join(HERE, "_ref", "libvisoref.so")

MATCH_DTYPE = np.dtype(
    [("u1p", "<f4"), ("v1p", "<f4"), ("i1p", "<i4"), ("u2p", "<f4"), ("v2p", "<f4"), ("i2p", "<i4"),
     ("u1c", "<f4"), ("v1c", "<f4"), ("i1c", "<i4"), ("u2c", "<f4"), ("v2c", "<f4"), ("i2c", "<i4")])
assert MATCH_DTYPE.itemsize == 48

PARAM_NAMES = ["nms_n", "nms_tau", "match_binsize", "match_radius", "match_disp_tolerance",
               "outlier_disp_tolerance", "outlier_flow_tolerance", "multi_stage", "half_resolution", "refinement"]
DEFAULT_PARAMS = dict(nms_n=3, nms_tau=50, match_binsize=50, match_radius=200, match_disp_tolerance=2,
                      outlier_disp_tolerance=5, outlier_flow_tolerance=5, multi_stage=1, half_resolution=1,
                      refinement=1, f=1.0, cu=0.0, cv=0.0, base=1.0)
FEATURE_SETS = {"1p1": 0, "2p1": 1, "1c1": 2, "2c1": 3, "1p2": 4, "2p2": 5, "1c2": 6, "2c2": 7}


class VoParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in PARAM_NAMES] + [(n, C.c_double) for n in ("f", "cu", "cv", "base")]


def make_params(**kw):
    d = dict(DEFAULT_PARAMS)
    d.update(kw)
    return d


def bpl16(w):
    return w + 15 - (w - 1) % 16


def pad_image(img):
    """(H,W) uint8 -> (H,bpl16(W)) uint8 with zero padding."""
    h, w = img.shape
    out = np.zeros((h, bpl16(w)), dtype=np.uint8)
    out[:, :w] = img
    return out


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(REF_SO)


_p_u8 = C.POINTER(C.c_uint8)
_p_i16 = C.POINTER(C.c_int16)
_p_i32 = C.POINTER(C.c_int32)
_p_f32 = C.POINTER(C.c_float)
_p_f64 = C.POINTER(C.c_double)


def _u8(a):
    return a.ctypes.data_as(_p_u8)


def _i16(a):
    return a.ctypes.data_as(_p_i16)


def _i32(a):
    return a.ctypes.data_as(_p_i32)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.vo_create.restype = C.c_void_p
        L.vo_create.argtypes = [C.POINTER(VoParams)]
        L.vo_destroy.argtypes = [C.c_void_p]
        L.vo_set_intrinsics.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.vo_push_back.argtypes = [C.c_void_p, _p_u8, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.vo_match_features.argtypes = [C.c_void_p, C.c_int32, _p_f64]
        L.vo_num_matches.argtypes = [C.c_void_p]
        L.vo_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_bucket_features.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float]
        L.vo_get_gain.argtypes = [C.c_void_p, _p_i32, C.c_int32]
        L.vo_get_gain.restype = C.c_float
        L.vo_stage_size.argtypes = [C.c_void_p, C.c_int32]
        L.vo_stage_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.vo_num_ranges.argtypes = [C.c_void_p]
        L.vo_get_ranges.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_num_features.argtypes = [C.c_void_p, C.c_int32]
        L.vo_get_features.argtypes = [C.c_void_p, C.c_int32, _p_i32]
        L.vo_get_gradients.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _p_u8, _p_u8]
        L.vo_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.vo_half_image.argtypes = [_p_u8, C.c_int32, C.c_int32, C.c_int32, _p_u8]
        L.vo_sobel5x5.argtypes = [_p_u8, _p_u8, _p_u8, C.c_int32, C.c_int32]
        L.vo_blob5x5.argtypes = [_p_u8, _p_i16, C.c_int32, C.c_int32]
        L.vo_checkerboard5x5.argtypes = [_p_u8, _p_i16, C.c_int32, C.c_int32]
        L.vo_nms.argtypes = [_p_i16, _p_i16] + [C.c_int32] * 5 + [_p_i32, C.c_int32]
        L.vo_delaunay.argtypes = [_p_f32, C.c_int32, _p_i32, C.c_int32]
        L.vo_remove_outliers.argtypes = [C.POINTER(VoParams), C.c_void_p, C.c_int32, C.c_int32]
        L.vo_ego_sampler_seed.argtypes = [C.c_uint32]
        L.vo_ego_sampler_state.restype = C.c_uint32
        L.vo_estimate_motion_stereo.argtypes = [C.c_void_p, C.c_int32, C.POINTER(VoEgoParams), _p_f64, _p_i32, _p_i32]
        L.vo_tr_vector_to_matrix.argtypes = [_p_f64, _p_f64]
        L.vo_stereo_create.restype = C.c_void_p
        L.vo_stereo_create.argtypes = [C.POINTER(VoParams), C.c_int32, C.c_double, C.c_double, C.POINTER(VoEgoParams)]
        L.vo_stereo_destroy.argtypes = [C.c_void_p]
        L.vo_stereo_process.argtypes = [C.c_void_p, _p_u8, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.vo_stereo_process_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.vo_stereo_get_motion.argtypes = [C.c_void_p, _p_f64]
        L.vo_stereo_tr_valid.argtypes = [C.c_void_p]
        L.vo_stereo_num_matches.argtypes = [C.c_void_p]
        L.vo_stereo_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_stereo_num_inliers.argtypes = [C.c_void_p]
        L.vo_stereo_get_inliers.argtypes = [C.c_void_p, _p_i32]
        L.vo_stereo_matcher.restype = C.c_void_p
        L.vo_stereo_matcher.argtypes = [C.c_void_p]
        L.vo_matrix_svd.argtypes = [_p_f64, C.c_int32, C.c_int32, _p_f64, _p_f64, _p_f64]
        L.vo_matrix_det.argtypes = [_p_f64, C.c_int32]
        L.vo_matrix_det.restype = C.c_double
        L.vo_mono_fundamental.argtypes = [C.c_void_p, _p_i32, C.c_int32, _p_f64]
        L.vo_estimate_motion_mono.argtypes = [C.c_void_p, C.c_int32, C.POINTER(VoMonoParams), _p_f64, _p_i32, _p_i32]
        L.vo_mono_create.restype = C.c_void_p
        L.vo_mono_create.argtypes = [C.POINTER(VoParams), C.c_int32, C.c_double, C.c_double, C.POINTER(VoMonoParams)]
        L.vo_mono_destroy.argtypes = [C.c_void_p]
        L.vo_mono_process.argtypes = [C.c_void_p, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.vo_mono_process_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.vo_mono_get_motion.argtypes = [C.c_void_p, _p_f64]
        L.vo_mono_num_matches.argtypes = [C.c_void_p]
        L.vo_mono_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_mono_num_inliers.argtypes = [C.c_void_p]
        L.vo_mono_get_inliers.argtypes = [C.c_void_p, _p_i32]
        _oracle = L
    return _oracle


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_matcher_create.restype = C.c_void_p
        L.ref_matcher_create.argtypes = [_p_i32, _p_f64]
        L.ref_matcher_destroy.argtypes = [C.c_void_p]
        L.ref_matcher_set_intrinsics.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.ref_matcher_push.argtypes = [C.c_void_p, _p_u8, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.ref_matcher_match.argtypes = [C.c_void_p, C.c_int32, _p_f64]
        L.ref_matcher_match_staged.argtypes = [C.c_void_p, C.c_int32, _p_f64]
        L.ref_matcher_stage_size.argtypes = [C.c_void_p, C.c_int32]
        L.ref_matcher_stage_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ref_matcher_num_ranges.argtypes = [C.c_void_p]
        L.ref_matcher_get_ranges.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_matcher_num_matches.argtypes = [C.c_void_p]
        L.ref_matcher_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_matcher_bucket.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float]
        L.ref_matcher_gain.argtypes = [C.c_void_p, _p_i32, C.c_int32]
        L.ref_matcher_gain.restype = C.c_float
        L.ref_matcher_num_features.argtypes = [C.c_void_p, C.c_int32]
        L.ref_matcher_get_features.argtypes = [C.c_void_p, C.c_int32, _p_i32]
        L.ref_matcher_get_gradients.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _p_u8, _p_u8]
        L.ref_sobel5x5.argtypes = [_p_u8, _p_u8, _p_u8, C.c_int32, C.c_int32]
        L.ref_blob5x5.argtypes = [_p_u8, _p_i16, C.c_int32, C.c_int32]
        L.ref_checkerboard5x5.argtypes = [_p_u8, _p_i16, C.c_int32, C.c_int32]
        L.ref_half_image.argtypes = [_p_u8, C.c_int32, C.c_int32, C.c_int32, _p_u8]
        L.ref_nms.argtypes = [_p_i16, _p_i16] + [C.c_int32] * 5 + [_p_i32, C.c_int32]
        L.ref_triangulate.argtypes = [_p_f32, C.c_int32, _p_i32, C.c_int32]
        L.ref_remove_outliers.argtypes = [_p_i32, _p_f64, C.c_void_p, C.c_int32, C.c_int32]
        L.ref_vo_stereo_create.restype = C.c_void_p
        L.ref_vo_stereo_create.argtypes = [_p_i32] + [C.c_double] * 4 + [C.c_int32, C.c_double, C.c_double]
        L.ref_vo_stereo_destroy.argtypes = [C.c_void_p]
        L.ref_vo_stereo_process.argtypes = [C.c_void_p, _p_u8, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            _p_f64, _p_f64]
        L.ref_vo_num_matches.argtypes = [C.c_void_p]
        L.ref_vo_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_vo_stereo_set_ransac.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_int32]
        L.ref_vo_process_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, _p_f64]
        L.ref_vo_num_bucketed.argtypes = [C.c_void_p]
        L.ref_vo_get_bucketed.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_vo_num_inliers.argtypes = [C.c_void_p]
        L.ref_vo_get_inliers.argtypes = [C.c_void_p, _p_i32]
        L.ref_vo_mono_create.restype = C.c_void_p
        L.ref_vo_mono_create.argtypes = [_p_i32] + [C.c_double] * 5 + [C.c_int32, C.c_double, C.c_double, C.c_int32,
                                                                         C.c_double, C.c_double]
        L.ref_vo_mono_destroy.argtypes = [C.c_void_p]
        L.ref_vo_mono_process.argtypes = [C.c_void_p, _p_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p_f64]
        L.ref_vo_mono_process_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, _p_f64]
        L.ref_vo_mono_num_bucketed.argtypes = [C.c_void_p]
        L.ref_vo_mono_get_bucketed.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_vo_mono_num_inliers.argtypes = [C.c_void_p]
        L.ref_vo_mono_get_inliers.argtypes = [C.c_void_p, _p_i32]
        L.ref_matrix_svd.argtypes = [_p_f64, C.c_int32, C.c_int32, _p_f64, _p_f64, _p_f64]
        L.ref_matrix_det.argtypes = [_p_f64, C.c_int32]
        L.ref_matrix_det.restype = C.c_double
        L.ref_mono_fundamental.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, _p_i32, C.c_int32, _p_f64]
        _ref = L
    return _ref


def _ip_dp(p):
    ip = np.array([p[n] for n in PARAM_NAMES], dtype=np.int32)
    dp = np.array([p["f"], p["cu"], p["cv"], p["base"]], dtype=np.float64)
    return ip, dp


def _vo_params(p):
    s = VoParams()
    for n in PARAM_NAMES:
        setattr(s, n, int(p[n]))
    for n in ("f", "cu", "cv", "base"):
        setattr(s, n, float(p[n]))
    return s


def _tr_ptr(Tr):
    if Tr is None:
        return None, None
    t = np.ascontiguousarray(np.asarray(Tr, dtype=np.float64).reshape(-1)[:12])
    return t, t.ctypes.data_as(_p_f64)


class CpuMatcher:
    """Common face over the oracle ("oracle") and the real reference ("ref")."""

    def __init__(self, kind="oracle", **params):
        self.kind = kind
        self.p = make_params(**params)
        if kind == "oracle":
            self.L = oracle_lib()
            sp = _vo_params(self.p)
            self.h = C.c_void_p(self.L.vo_create(C.byref(sp)))
        else:
            self.L = ref_lib()
            ip, dp = _ip_dp(self.p)
            self.h = C.c_void_p(self.L.ref_matcher_create(_i32(ip), dp.ctypes.data_as(_p_f64)))

    def close(self):
        if self.h:
            (self.L.vo_destroy if self.kind == "oracle" else self.L.ref_matcher_destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_intrinsics(self, f, cu, cv, base):
        (self.L.vo_set_intrinsics if self.kind == "oracle" else self.L.ref_matcher_set_intrinsics)(self.h, f, cu, cv, base)

    def push_back(self, I1, I2=None, replace=False):
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        h, w = I1.shape
        p2 = None
        if I2 is not None:
            I2 = np.ascontiguousarray(I2, dtype=np.uint8)
            assert I2.shape == I1.shape
            p2 = _u8(I2)
        fn = self.L.vo_push_back if self.kind == "oracle" else self.L.ref_matcher_push
        return fn(self.h, _u8(I1), p2, w, h, w, int(replace))

    def match(self, method, Tr=None):
        """runs matchFeatures with stage capture; returns True if matching ran."""
        t, tp = _tr_ptr(Tr)
        if self.kind == "oracle":
            return self.L.vo_match_features(self.h, method, tp) == 1
        return self.L.ref_matcher_match_staged(self.h, method, tp) == 1

    def matches(self):
        n = (self.L.vo_num_matches if self.kind == "oracle" else self.L.ref_matcher_num_matches)(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            (self.L.vo_get_matches if self.kind == "oracle" else self.L.ref_matcher_get_matches)(self.h, out.ctypes.data)
        return out

    def stage(self, s):
        n = (self.L.vo_stage_size if self.kind == "oracle" else self.L.ref_matcher_stage_size)(self.h, s)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            (self.L.vo_stage_get if self.kind == "oracle" else self.L.ref_matcher_stage_get)(self.h, s, out.ctypes.data)
        return out

    def ranges(self):
        n = (self.L.vo_num_ranges if self.kind == "oracle" else self.L.ref_matcher_num_ranges)(self.h)
        out = np.zeros((n, 4, 4), dtype=np.float32)  # [bin][u_min,u_max,v_min,v_max][stage]
        if n:
            (self.L.vo_get_ranges if self.kind == "oracle" else self.L.ref_matcher_get_ranges)(self.h, out.ctypes.data)
        return out

    def features(self, which):
        w = FEATURE_SETS[which] if isinstance(which, str) else which
        n = (self.L.vo_num_features if self.kind == "oracle" else self.L.ref_matcher_num_features)(self.h, w)
        out = np.zeros((n, 12), dtype=np.int32)
        if n:
            (self.L.vo_get_features if self.kind == "oracle" else self.L.ref_matcher_get_features)(self.h, w, _i32(out))
        return out

    def gradients(self, which, full, shape):
        du = np.zeros(shape, dtype=np.uint8)
        dv = np.zeros(shape, dtype=np.uint8)
        fn = self.L.vo_get_gradients if self.kind == "oracle" else self.L.ref_matcher_get_gradients
        n = fn(self.h, which, int(full), _u8(du), _u8(dv))
        assert n == 0 or n == du.size, (n, du.size)
        return (du, dv) if n else (None, None)

    def bucket(self, max_features, bw, bh):
        (self.L.vo_bucket_features if self.kind == "oracle" else self.L.ref_matcher_bucket)(self.h, max_features, bw, bh)

    def gain(self, inliers):
        a = np.ascontiguousarray(inliers, dtype=np.int32)
        return float((self.L.vo_get_gain if self.kind == "oracle" else self.L.ref_matcher_gain)(self.h, _i32(a), len(a)))

    def counters(self):
        assert self.kind == "oracle"
        c = (C.c_int64 * 8)()
        self.L.vo_get_counters(self.h, c)
        return dict(zip(("Q", "C", "S", "M", "M_out", "Q1", "C1", "S1"), list(c)))


# ---- free-standing stages -------------------------------------------------

def half_image(kind, img_padded, w):
    h, bpl = img_padded.shape
    out = np.zeros((h // 2, bpl16(w // 2)), dtype=np.uint8)
    fn = oracle_lib().vo_half_image if kind == "oracle" else ref_lib().ref_half_image
    fn(_u8(np.ascontiguousarray(img_padded)), w, h, bpl, _u8(out))
    return out


def sobel5x5(kind, img_padded):
    h, bpl = img_padded.shape
    du = np.zeros((h, bpl), dtype=np.uint8)
    dv = np.zeros((h, bpl), dtype=np.uint8)
    fn = oracle_lib().vo_sobel5x5 if kind == "oracle" else ref_lib().ref_sobel5x5
    fn(_u8(np.ascontiguousarray(img_padded)), _u8(du), _u8(dv), bpl, h)
    return du, dv


def blob5x5(kind, img_padded):
    h, bpl = img_padded.shape
    out = np.zeros((h, bpl), dtype=np.int16)
    fn = oracle_lib().vo_blob5x5 if kind == "oracle" else ref_lib().ref_blob5x5
    fn(_u8(np.ascontiguousarray(img_padded)), _i16(out), bpl, h)
    return out


def checkerboard5x5(kind, img_padded):
    h, bpl = img_padded.shape
    out = np.zeros((h, bpl), dtype=np.int16)
    fn = oracle_lib().vo_checkerboard5x5 if kind == "oracle" else ref_lib().ref_checkerboard5x5
    fn(_u8(np.ascontiguousarray(img_padded)), _i16(out), bpl, h)
    return out


def nms(kind, f1, f2, w, n, tau):
    h, bpl = f1.shape
    cap = 4 * (w // (n + 1) + 1) * (h // (n + 1) + 1) + 16
    out = np.zeros((cap, 4), dtype=np.int32)
    fn = oracle_lib().vo_nms if kind == "oracle" else ref_lib().ref_nms
    k = fn(_i16(np.ascontiguousarray(f1)), _i16(np.ascontiguousarray(f2)), w, h, bpl, n, tau, _i32(out), cap)
    assert k <= cap
    return out[:k]


def delaunay(kind, pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    n = len(pts)
    cap = 2 * n + 16
    tris = np.zeros((cap, 3), dtype=np.int32)
    fn = oracle_lib().vo_delaunay if kind == "oracle" else ref_lib().ref_triangulate
    k = fn(pts.ctypes.data_as(_p_f32), n, _i32(tris), cap)
    assert k <= cap
    return tris[:k]


def remove_outliers(kind, matches, method, **params):
    p = make_params(**params)
    m = np.ascontiguousarray(matches.copy())
    if kind == "oracle":
        sp = _vo_params(p)
        k = oracle_lib().vo_remove_outliers(C.byref(sp), m.ctypes.data, len(m), method)
    else:
        ip, dp = _ip_dp(p)
        k = ref_lib().ref_remove_outliers(_i32(ip), dp.ctypes.data_as(_p_f64), m.ctypes.data, len(m), method)
    return m[:k]


class RefStereoVO:
    """VisualOdometryStereo of the reference (Tr_delta fixtures)."""

    def __init__(self, f, cu, cv, base, bucket=(2, 50.0, 50.0), ransac_iters=200, inlier_threshold=2.0,
                 reweighting=True, **params):
        self.L = ref_lib()
        p = make_params(**params)
        ip, _ = _ip_dp(p)
        self.h = C.c_void_p(self.L.ref_vo_stereo_create(_i32(ip), f, cu, cv, base, bucket[0], bucket[1], bucket[2]))
        self.L.ref_vo_stereo_set_ransac(self.h, ransac_iters, inlier_threshold, int(reweighting))

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
        tout = np.zeros(16)
        ok = self.L.ref_vo_process_matches(self.h, m.ctypes.data, len(m), tout.ctypes.data_as(_p_f64))
        return bool(ok), tout.reshape(4, 4)

    def bucketed(self):
        n = self.L.ref_vo_num_bucketed(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            self.L.ref_vo_get_bucketed(self.h, out.ctypes.data)
        return out

    def inliers(self):
        n = self.L.ref_vo_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            self.L.ref_vo_get_inliers(self.h, _i32(out))
        return out

    def process(self, I1, I2, replace=False):
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        I2 = np.ascontiguousarray(I2, dtype=np.uint8)
        h, w = I1.shape
        tin = np.zeros(16)
        tout = np.zeros(16)
        rc = self.L.ref_vo_stereo_process(self.h, _u8(I1), _u8(I2), w, h, w, int(replace),
                                          tin.ctypes.data_as(_p_f64), tout.ctypes.data_as(_p_f64))
        return bool(rc & 1), bool(rc & 2), tin.reshape(4, 4), tout.reshape(4, 4)

    def matches(self):
        n = self.L.ref_vo_num_matches(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            self.L.ref_vo_get_matches(self.h, out.ctypes.data)
        return out

    def close(self):
        if self.h:
            self.L.ref_vo_stereo_destroy(self.h)
            self.h = None


class VoEgoParams(C.Structure):
    _fields_ = [("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double), ("base", C.c_double),
                ("ransac_iters", C.c_int32), ("inlier_threshold", C.c_double), ("reweighting", C.c_int32)]


def ego_params(f, cu, cv, base, ransac_iters=200, inlier_threshold=2.0, reweighting=True):
    e = VoEgoParams()
    e.f, e.cu, e.cv, e.base = float(f), float(cu), float(cv), float(base)
    e.ransac_iters, e.inlier_threshold, e.reweighting = int(ransac_iters), float(inlier_threshold), int(reweighting)
    return e


def oracle_sampler_seed(s=71):
    oracle_lib().vo_ego_sampler_seed(s)


def oracle_estimate_motion(m, ep):
    """vo_estimate_motion_stereo -> (rc, tr6, inliers)"""
    L = oracle_lib()
    m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
    tr = np.zeros(6)
    inl = np.zeros(max(len(m), 1), dtype=np.int32)
    n = C.c_int32(-1)
    rc = L.vo_estimate_motion_stereo(m.ctypes.data, len(m), C.byref(ep), tr.ctypes.data_as(_p_f64), _i32(inl),
                                     C.byref(n))
    return rc, tr, inl[:max(n.value, 0)].copy()


class OracleStereoVO:
    """The oracle's VisualOdometryStereo; same face as RefStereoVO."""

    def __init__(self, f, cu, cv, base, bucket=(2, 50.0, 50.0), ransac_iters=200, inlier_threshold=2.0,
                 reweighting=True, **params):
        self.L = oracle_lib()
        sp = _vo_params(make_params(**params))
        ep = ego_params(f, cu, cv, base, ransac_iters, inlier_threshold, reweighting)
        self.h = C.c_void_p(self.L.vo_stereo_create(C.byref(sp), bucket[0], bucket[1], bucket[2], C.byref(ep)))

    def motion(self):
        t = np.zeros(16)
        self.L.vo_stereo_get_motion(self.h, t.ctypes.data_as(_p_f64))
        return t.reshape(4, 4)

    def process(self, I1, I2, replace=False):
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        I2 = np.ascontiguousarray(I2, dtype=np.uint8)
        h, w = I1.shape
        valid = bool(self.L.vo_stereo_tr_valid(self.h))
        tin = self.motion()
        ok = self.L.vo_stereo_process(self.h, _u8(I1), _u8(I2), w, h, w, int(replace))
        return bool(ok), valid, tin, self.motion()

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
        ok = self.L.vo_stereo_process_matches(self.h, m.ctypes.data, len(m))
        return bool(ok), self.motion()

    def bucketed(self):
        n = self.L.vo_stereo_num_matches(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            self.L.vo_stereo_get_matches(self.h, out.ctypes.data)
        return out

    def inliers(self):
        n = self.L.vo_stereo_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            self.L.vo_stereo_get_inliers(self.h, _i32(out))
        return out

    def close(self):
        if self.h:
            self.L.vo_stereo_destroy(self.h)
            self.h = None


MONO_DEFAULTS = dict(height=1.0, pitch=0.0, ransac_iters=2000, inlier_threshold=0.00001, motion_threshold=100.0)


class RefMonoVO:
    """VisualOdometryMono of the reference (viso/viso_mono.h:28-90)."""

    def __init__(self, f, cu, cv, bucket=(2, 50.0, 50.0), **kw):
        self.L = ref_lib()
        mono = dict(MONO_DEFAULTS)
        mono.update({k: kw.pop(k) for k in list(kw) if k in MONO_DEFAULTS})
        p = make_params(**kw)
        ip, _ = _ip_dp(p)
        self.h = C.c_void_p(self.L.ref_vo_mono_create(_i32(ip), f, cu, cv, mono["height"], mono["pitch"],
                                                      mono["ransac_iters"], mono["inlier_threshold"],
                                                      mono["motion_threshold"], bucket[0], bucket[1], bucket[2]))

    def process(self, I, replace=False):
        I = np.ascontiguousarray(I, dtype=np.uint8)
        h, w = I.shape
        t = np.zeros(16)
        ok = self.L.ref_vo_mono_process(self.h, _u8(I), w, h, w, int(replace), t.ctypes.data_as(_p_f64))
        return bool(ok), t.reshape(4, 4)

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
        t = np.zeros(16)
        ok = self.L.ref_vo_mono_process_matches(self.h, m.ctypes.data, len(m), t.ctypes.data_as(_p_f64))
        return bool(ok), t.reshape(4, 4)

    def bucketed(self):
        n = self.L.ref_vo_mono_num_bucketed(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            self.L.ref_vo_mono_get_bucketed(self.h, out.ctypes.data)
        return out

    def inliers(self):
        n = self.L.ref_vo_mono_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            self.L.ref_vo_mono_get_inliers(self.h, _i32(out))
        return out

    def fundamental(self, m, active):
        m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
        a = np.ascontiguousarray(active, dtype=np.int32)
        F = np.zeros(9)
        self.L.ref_mono_fundamental(self.h, m.ctypes.data, len(m), _i32(a), len(a), F.ctypes.data_as(_p_f64))
        return F.reshape(3, 3)

    def close(self):
        if self.h:
            self.L.ref_vo_mono_destroy(self.h)
            self.h = None


def ref_svd(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    U, W, V = np.zeros((m, m)), np.zeros(min(m, n)), np.zeros((n, n))
    ref_lib().ref_matrix_svd(A.ctypes.data_as(_p_f64), m, n, U.ctypes.data_as(_p_f64), W.ctypes.data_as(_p_f64),
                             V.ctypes.data_as(_p_f64))
    return U, W, V


def ref_det(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    return float(ref_lib().ref_matrix_det(A.ctypes.data_as(_p_f64), A.shape[0]))


class VoMonoParams(C.Structure):
    _fields_ = [("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double), ("height", C.c_double),
                ("pitch", C.c_double), ("ransac_iters", C.c_int32), ("inlier_threshold", C.c_double),
                ("motion_threshold", C.c_double)]


def mono_params(f, cu, cv, **kw):
    d = dict(MONO_DEFAULTS)
    d.update(kw)
    e = VoMonoParams()
    e.f, e.cu, e.cv = float(f), float(cu), float(cv)
    e.height, e.pitch, e.ransac_iters = float(d["height"]), float(d["pitch"]), int(d["ransac_iters"])
    e.inlier_threshold, e.motion_threshold = float(d["inlier_threshold"]), float(d["motion_threshold"])
    return e


def oracle_svd(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    U, W, V = np.zeros((m, m)), np.zeros(min(m, n)), np.zeros((n, n))
    oracle_lib().vo_matrix_svd(A.ctypes.data_as(_p_f64), m, n, U.ctypes.data_as(_p_f64), W.ctypes.data_as(_p_f64),
                               V.ctypes.data_as(_p_f64))
    return U, W, V


def oracle_det(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    return float(oracle_lib().vo_matrix_det(A.ctypes.data_as(_p_f64), A.shape[0]))


def oracle_fundamental(m, active):
    m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
    a = np.ascontiguousarray(active, dtype=np.int32)
    F = np.zeros(9)
    oracle_lib().vo_mono_fundamental(m.ctypes.data, _i32(a), len(a), F.ctypes.data_as(_p_f64))
    return F.reshape(3, 3)


class OracleMonoVO:
    """The oracle's VisualOdometryMono; same face as RefMonoVO."""

    def __init__(self, f, cu, cv, bucket=(2, 50.0, 50.0), **kw):
        self.L = oracle_lib()
        mono = {k: kw.pop(k) for k in list(kw) if k in MONO_DEFAULTS}
        sp = _vo_params(make_params(**kw))
        ep = mono_params(f, cu, cv, **mono)
        self.h = C.c_void_p(self.L.vo_mono_create(C.byref(sp), bucket[0], bucket[1], bucket[2], C.byref(ep)))

    def motion(self):
        t = np.zeros(16)
        self.L.vo_mono_get_motion(self.h, t.ctypes.data_as(_p_f64))
        return t.reshape(4, 4)

    def process(self, I, replace=False):
        I = np.ascontiguousarray(I, dtype=np.uint8)
        h, w = I.shape
        ok = self.L.vo_mono_process(self.h, _u8(I), w, h, w, int(replace))
        return bool(ok), self.motion()

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=MATCH_DTYPE)
        ok = self.L.vo_mono_process_matches(self.h, m.ctypes.data, len(m))
        return bool(ok), self.motion()

    def bucketed(self):
        n = self.L.vo_mono_num_matches(self.h)
        out = np.zeros(n, dtype=MATCH_DTYPE)
        if n:
            self.L.vo_mono_get_matches(self.h, out.ctypes.data)
        return out

    def inliers(self):
        n = self.L.vo_mono_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            self.L.vo_mono_get_inliers(self.h, _i32(out))
        return out

    def close(self):
        if self.h:
            self.L.vo_mono_destroy(self.h)
            self.h = None
