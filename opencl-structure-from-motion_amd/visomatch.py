"""ctypes binding of libvisomatch.so (include/visomatch.h) -- the Python mirror of the
reference's `class Matcher` (viso/matcher.h:37-136): pushBack / matchFeatures / getMatches /
bucketFeatures / getGain / setIntrinsics, plus the stage-level views the parity tests use.

The library is the HIP build for gfx950; there is no CPU path.  Importing works anywhere (the
CPU test-suite checks the exported symbols), creating a Matcher needs a GPU and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# The look-ahead path uses four streams (the handle's and three side streams the Delaunay chains rotate over).  The HIP
# runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: two of ours would share one and serialise), and
# with SIX or more hardware queues in a process every kernel's workgroup dispatch runs at a half or a quarter of its rate
# (DESIGN_HISTORY.md 6c), so: five - ours and the null stream's.  Read when the runtime initialises, so it only helps if nothing in
# this process has touched the GPU yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
LIB_PATH = os.environ.get("VSM_LIB_PATH") or os.path.join(HERE, "libvisomatch.so")  # override: kernel experiments

P_MATCH = np.dtype(
    [("u1p", "<f4"), ("v1p", "<f4"), ("i1p", "<i4"), ("u2p", "<f4"), ("v2p", "<f4"), ("i2p", "<i4"),
     ("u1c", "<f4"), ("v1c", "<f4"), ("i1c", "<i4"), ("u2c", "<f4"), ("v2c", "<f4"), ("i2c", "<i4")])
assert P_MATCH.itemsize == 48

INT_PARAMS = ["nms_n", "nms_tau", "match_binsize", "match_radius", "match_disp_tolerance",
              "outlier_disp_tolerance", "outlier_flow_tolerance", "multi_stage", "half_resolution", "refinement"]
FEATURE_SETS = {"1p1": 0, "2p1": 1, "1c1": 2, "2c1": 3, "1p2": 4, "2p2": 5, "1c2": 6, "2c2": 7}

# every symbol include/visomatch.h declares
EXPORTS = ["vsm_default_params", "vsm_create", "vsm_destroy", "vsm_set_intrinsics", "vsm_push_back",
           "vsm_push_back_device", "vsm_wait_for_stream", "vsm_match", "vsm_num_matches", "vsm_get_matches", "vsm_bucket", "vsm_gain",
           "vsm_num_features", "vsm_get_features", "vsm_set_stage_capture", "vsm_stage_size", "vsm_stage_get",
           "vsm_num_ranges", "vsm_get_ranges", "vsm_get_gradients", "vsm_get_filter_responses", "vsm_get_counters",
           "vsm_get_timings", "vsm_set_profiling", "vsm_num_kernels", "vsm_kernel_name", "vsm_get_kernel_stats",
           "vsm_host_delaunay", "vsm_host_delaunay_split", "vsm_debug_delaunay_gpu", "vsm_debug_dc_bench", "vsm_host_ties", "vsm_debug_ties_gpu", "vsm_host_outliers_and_prior", "vsm_host_outliers_and_prior_threads", "vsm_debug_dc2", "vsm_debug_dc2_band_factor", "vsm_local_cpus", "vsm_forkjoin_cpus", "vsm_device_pool_stats", "vsm_device_pool_trim", "vsm_sequence_run", "vsm_sequence_num_matches", "vsm_sequence_get_matches",
           "vsm_sequence_get_timings", "vsm_sequence_path", "vsm_set_option", "vsm_version", "vsm_host_register", "vsm_host_unregister",
           "vsm_multi_create", "vsm_multi_destroy", "vsm_multi_process", "vsm_multi_num_sequences", "vsm_multi_get_motion",
           "vsm_multi_motion_valid", "vsm_multi_num_matches", "vsm_multi_get_matches", "vsm_multi_num_inliers", "vsm_multi_get_inliers",
           "vsm_multi_get_timings",
           "vsm_vo_stereo_default_params", "vsm_vo_stereo_create", "vsm_vo_stereo_destroy", "vsm_vo_stereo_process",
           "vsm_vo_stereo_process_device", "vsm_vo_stereo_process_matches", "vsm_vo_stereo_get_motion",
           "vsm_vo_stereo_motion_valid", "vsm_vo_stereo_num_matches", "vsm_vo_stereo_get_matches",
           "vsm_vo_stereo_num_inliers", "vsm_vo_stereo_get_inliers", "vsm_vo_stereo_gain", "vsm_vo_stereo_matcher",
           "vsm_vo_stereo_get_timings", "vsm_vo_sampler_seed", "vsm_host_estimate_motion_stereo",
           "vsm_vo_mono_default_params", "vsm_vo_mono_create", "vsm_vo_mono_destroy", "vsm_vo_mono_process",
           "vsm_vo_mono_process_device", "vsm_vo_mono_process_matches", "vsm_vo_mono_get_motion",
           "vsm_vo_mono_motion_valid", "vsm_vo_mono_num_matches", "vsm_vo_mono_get_matches", "vsm_vo_mono_num_inliers",
           "vsm_vo_mono_get_inliers", "vsm_vo_mono_gain", "vsm_vo_mono_matcher", "vsm_vo_mono_get_timings",
           "vsm_vo_mono_device_svd",
           "vsm_host_estimate_motion_mono"]


class VsmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in INT_PARAMS] + [(n, C.c_double) for n in ("f", "cu", "cv", "base")]


class VsmVoStereoParams(C.Structure):
    _fields_ = [("match", VsmParams), ("bucket_max_features", C.c_int32), ("bucket_width", C.c_double),
                ("bucket_height", C.c_double), ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double),
                ("base", C.c_double), ("ransac_iters", C.c_int32), ("inlier_threshold", C.c_double),
                ("reweighting", C.c_int32)]


class VsmVoMonoParams(C.Structure):
    _fields_ = [("match", VsmParams), ("bucket_max_features", C.c_int32), ("bucket_width", C.c_double),
                ("bucket_height", C.c_double), ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double),
                ("height", C.c_double), ("pitch", C.c_double), ("ransac_iters", C.c_int32),
                ("inlier_threshold", C.c_double), ("motion_threshold", C.c_double)]


def host_register(arr):
    """page-lock a (contiguous) numpy array for DMA (vsm_host_register = hipHostRegister); True on success.  Pair with
    Matcher.set_option("seq_host_pinned", 1) and host_unregister(arr) before the array is freed."""
    assert arr.flags["C_CONTIGUOUS"]
    return lib().vsm_host_register(arr.ctypes.data_as(C.c_void_p), arr.nbytes) == 0


def host_unregister(arr):
    return lib().vsm_host_unregister(arr.ctypes.data_as(C.c_void_p)) == 0


def device_pool_stats():
    """(blocks in the cache, their bytes, large blocks in use): the device memory closed handles left for the next one"""
    out = (C.c_int64 * 3)()
    lib().vsm_device_pool_stats(out)
    return int(out[0]), int(out[1]), int(out[2])


def device_pool_trim():
    lib().vsm_device_pool_trim()


class VisoMatchError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VisoMatchError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP
        # runtimes in one process cannot both own the GPU, so when torch is installed let it load
        # its runtime first; libvisomatch.so then binds to that copy.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch-less environments use /opt/rocm's runtime
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        L.vsm_version.restype = C.c_char_p
        L.vsm_default_params.argtypes = [C.POINTER(VsmParams)]
        L.vsm_create.restype = vp
        L.vsm_create.argtypes = [C.POINTER(VsmParams)]
        L.vsm_destroy.argtypes = [vp]
        L.vsm_set_intrinsics.argtypes = [vp] + [C.c_double] * 4
        L.vsm_push_back.argtypes = [vp, vp, vp, i32, i32, i32, C.c_int]
        L.vsm_push_back_device.argtypes = [vp, vp, vp, i32, i32, i32, C.c_int]
        L.vsm_wait_for_stream.argtypes = [vp, vp]
        L.vsm_match.argtypes = [vp, i32, vp]
        L.vsm_num_matches.argtypes = [vp]
        L.vsm_get_matches.argtypes = [vp, vp, i32]
        L.vsm_bucket.argtypes = [vp, i32, f32, f32]
        L.vsm_gain.argtypes = [vp, vp, i32]
        L.vsm_gain.restype = f32
        L.vsm_num_features.argtypes = [vp, i32]
        L.vsm_get_features.argtypes = [vp, i32, vp, i32]
        L.vsm_set_stage_capture.argtypes = [vp, C.c_int]
        L.vsm_stage_size.argtypes = [vp, i32]
        L.vsm_stage_get.argtypes = [vp, i32, vp, i32]
        L.vsm_num_ranges.argtypes = [vp]
        L.vsm_get_ranges.argtypes = [vp, vp, i32]
        L.vsm_get_gradients.argtypes = [vp, i32, i32, vp, vp]
        L.vsm_get_filter_responses.argtypes = [vp, vp, vp]
        L.vsm_get_counters.argtypes = [vp, vp]
        L.vsm_get_timings.argtypes = [vp, vp]
        L.vsm_set_profiling.argtypes = [vp, C.c_int]
        L.vsm_kernel_name.restype = C.c_char_p
        L.vsm_kernel_name.argtypes = [i32]
        L.vsm_get_kernel_stats.argtypes = [vp, vp, vp]
        L.vsm_host_delaunay.argtypes = [vp, vp, i32, vp, i32, i32]
        L.vsm_host_delaunay_split.argtypes = [vp, vp, i32, vp, i32, i32, i32]
        L.vsm_debug_delaunay_gpu.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32]
        L.vsm_debug_dc_bench.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32]
        L.vsm_debug_dc_bench.restype = C.c_double
        L.vsm_host_ties.argtypes = [vp, vp, i32, vp, i32]
        L.vsm_debug_ties_gpu.argtypes = [vp, vp, i32, vp, i32, vp]
        L.vsm_host_outliers_and_prior.argtypes = [C.POINTER(VsmParams), vp, i32, i32, vp, i32, vp, i32, i32]
        L.vsm_host_outliers_and_prior_threads.argtypes = [C.POINTER(VsmParams), vp, i32, i32, vp, i32, vp, i32, i32, i32]
        L.vsm_debug_dc2.argtypes = [C.POINTER(VsmParams), vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, vp]
        L.vsm_debug_dc2_band_factor.argtypes = [i32]
        L.vsm_local_cpus.argtypes = [vp, i32]
        L.vsm_forkjoin_cpus.argtypes = [vp, i32]
        L.vsm_debug_dc2_band_factor.restype = None
        L.vsm_sequence_run.argtypes = [vp, vp, vp, C.c_int64, C.c_int, i32, i32, i32, i32, i32, vp, vp]
        L.vsm_sequence_num_matches.argtypes = [vp, i32]
        L.vsm_sequence_get_matches.argtypes = [vp, i32, vp, i32]
        L.vsm_sequence_get_timings.argtypes = [vp, vp]
        L.vsm_sequence_path.argtypes = [vp]
        L.vsm_set_option.argtypes = [vp, C.c_char_p, i32]
        L.vsm_host_register.argtypes = [vp, C.c_uint64]
        L.vsm_host_unregister.argtypes = [vp]
        L.vsm_device_pool_stats.argtypes = [vp]
        L.vsm_device_pool_stats.restype = None
        L.vsm_device_pool_trim.argtypes = []
        L.vsm_device_pool_trim.restype = None
        L.vsm_multi_create.restype = vp
        L.vsm_multi_create.argtypes = [C.POINTER(VsmVoStereoParams), i32]
        L.vsm_multi_destroy.argtypes = [vp]
        L.vsm_multi_process.argtypes = [vp, vp, vp, C.c_int64, C.c_int, i32, i32, i32, vp]
        L.vsm_multi_num_sequences.argtypes = [vp]
        L.vsm_multi_get_motion.argtypes = [vp, i32, vp]
        L.vsm_multi_motion_valid.argtypes = [vp, i32]
        L.vsm_multi_num_matches.argtypes = [vp, i32, C.c_int]
        L.vsm_multi_get_matches.argtypes = [vp, i32, C.c_int, vp, i32]
        L.vsm_multi_num_inliers.argtypes = [vp, i32]
        L.vsm_multi_get_inliers.argtypes = [vp, i32, vp, i32]
        L.vsm_multi_get_timings.argtypes = [vp, vp]
        vop = C.POINTER(VsmVoStereoParams)
        L.vsm_vo_stereo_default_params.argtypes = [vop]
        L.vsm_vo_stereo_create.restype = vp
        L.vsm_vo_stereo_create.argtypes = [vop]
        L.vsm_vo_stereo_destroy.argtypes = [vp]
        L.vsm_vo_stereo_process.argtypes = [vp, vp, vp, i32, i32, i32, C.c_int]
        L.vsm_vo_stereo_process_device.argtypes = [vp, vp, vp, i32, i32, i32, C.c_int]
        L.vsm_vo_stereo_process_matches.argtypes = [vp, vp, i32]
        L.vsm_vo_stereo_get_motion.argtypes = [vp, vp]
        L.vsm_vo_stereo_motion_valid.argtypes = [vp]
        L.vsm_vo_stereo_num_matches.argtypes = [vp]
        L.vsm_vo_stereo_get_matches.argtypes = [vp, vp, i32]
        L.vsm_vo_stereo_num_inliers.argtypes = [vp]
        L.vsm_vo_stereo_get_inliers.argtypes = [vp, vp, i32]
        L.vsm_vo_stereo_gain.argtypes = [vp, vp, i32]
        L.vsm_vo_stereo_gain.restype = f32
        L.vsm_vo_stereo_matcher.restype = vp
        L.vsm_vo_stereo_matcher.argtypes = [vp]
        L.vsm_vo_stereo_get_timings.argtypes = [vp, vp]
        L.vsm_vo_sampler_seed.argtypes = [C.c_uint32]
        L.vsm_host_estimate_motion_stereo.argtypes = [vop, vp, i32, i32, vp, vp, vp, vp]
        mop = C.POINTER(VsmVoMonoParams)
        L.vsm_vo_mono_default_params.argtypes = [mop]
        L.vsm_vo_mono_create.restype = vp
        L.vsm_vo_mono_create.argtypes = [mop]
        L.vsm_vo_mono_destroy.argtypes = [vp]
        L.vsm_vo_mono_process.argtypes = [vp, vp, i32, i32, i32, C.c_int]
        L.vsm_vo_mono_process_device.argtypes = [vp, vp, i32, i32, i32, C.c_int]
        L.vsm_vo_mono_process_matches.argtypes = [vp, vp, i32]
        L.vsm_vo_mono_get_motion.argtypes = [vp, vp]
        L.vsm_vo_mono_motion_valid.argtypes = [vp]
        L.vsm_vo_mono_num_matches.argtypes = [vp]
        L.vsm_vo_mono_get_matches.argtypes = [vp, vp, i32]
        L.vsm_vo_mono_num_inliers.argtypes = [vp]
        L.vsm_vo_mono_get_inliers.argtypes = [vp, vp, i32]
        L.vsm_vo_mono_gain.argtypes = [vp, vp, i32]
        L.vsm_vo_mono_gain.restype = f32
        L.vsm_vo_mono_matcher.restype = vp
        L.vsm_vo_mono_matcher.argtypes = [vp]
        L.vsm_vo_mono_get_timings.argtypes = [vp, vp]
        L.vsm_vo_mono_device_svd.argtypes = [vp]
        L.vsm_host_estimate_motion_mono.argtypes = [mop, vp, i32, i32, vp, vp, vp, vp]
        _lib = L
    return _lib


MONO_DEFAULTS = dict(height=1.0, pitch=0.0, ransac_iters=2000, inlier_threshold=0.00001, motion_threshold=100.0)


def vo_mono_params(f=1.0, cu=0.0, cv=0.0, bucket=(2, 50.0, 50.0), **kw):
    """VisualOdometryMono::parameters (viso/viso_mono.h:33-46, viso/viso.h:33-60) as the C struct"""
    p = VsmVoMonoParams()
    lib().vsm_vo_mono_default_params(C.byref(p))
    for k in list(kw):
        if k in MONO_DEFAULTS:
            v = kw.pop(k)
            setattr(p, k, int(v) if k == "ransac_iters" else float(v))
    for k, v in kw.items():
        if not hasattr(p.match, k):
            raise TypeError(f"unknown matcher parameter {k}")
        setattr(p.match, k, v)
    p.bucket_max_features, p.bucket_width, p.bucket_height = int(bucket[0]), float(bucket[1]), float(bucket[2])
    p.f, p.cu, p.cv = float(f), float(cu), float(cv)
    return p


def host_estimate_motion_mono(matches, params, threads=1):
    """mono egomotion of the product on a given flow-match list, host code only (no GPU)
    -> (rc, tr6, T 4x4, inliers or None)"""
    m = np.ascontiguousarray(matches, dtype=P_MATCH)
    tr = np.zeros(6)
    T = np.zeros(16)
    inl = np.zeros(max(len(m), 1), dtype=np.int32)
    n = C.c_int32(0)
    rc = lib().vsm_host_estimate_motion_mono(C.byref(params), m.ctypes.data_as(C.c_void_p), len(m), threads,
                                             tr.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                             inl.ctypes.data_as(C.c_void_p), C.cast(C.byref(n), C.c_void_p))
    return rc, tr, T.reshape(4, 4), (inl[: n.value].copy() if rc >= 0 else None)


def vo_stereo_params(f=1.0, cu=0.0, cv=0.0, base=1.0, bucket=(2, 50.0, 50.0), ransac_iters=200, inlier_threshold=2.0,
                     reweighting=True, **match):
    """VisualOdometryStereo::parameters (viso/viso_stereo.h:33-44, viso/viso.h:33-60) as the C struct"""
    p = VsmVoStereoParams()
    lib().vsm_vo_stereo_default_params(C.byref(p))
    for k, v in match.items():
        if not hasattr(p.match, k):
            raise TypeError(f"unknown matcher parameter {k}")
        setattr(p.match, k, v)
    p.bucket_max_features, p.bucket_width, p.bucket_height = int(bucket[0]), float(bucket[1]), float(bucket[2])
    p.f, p.cu, p.cv, p.base = float(f), float(cu), float(cv), float(base)
    p.ransac_iters, p.inlier_threshold, p.reweighting = int(ransac_iters), float(inlier_threshold), int(reweighting)
    return p


def vo_sampler_seed(seed=71):
    """re-seed the process-wide RANSAC sampler (viso/viso.cpp:93 seeds it with 71 once per process)"""
    lib().vsm_vo_sampler_seed(seed)


def host_estimate_motion_stereo(matches, params, threads=1):
    """egomotion solver of the product on a given match list (host code; no GPU needed)
    -> (rc, tr6, T 4x4, inliers or None); rc 1 ok, 0 failed, -1 fewer than 6 matches (inliers None)"""
    m = np.ascontiguousarray(matches, dtype=P_MATCH)
    tr = np.zeros(6)
    T = np.zeros(16)
    inl = np.zeros(max(len(m), 1), dtype=np.int32)
    n = C.c_int32(0)
    rc = lib().vsm_host_estimate_motion_stereo(C.byref(params), m.ctypes.data_as(C.c_void_p), len(m), threads,
                                               tr.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p),
                                               inl.ctypes.data_as(C.c_void_p), C.cast(C.byref(n), C.c_void_p))
    return rc, tr, T.reshape(4, 4), (inl[: n.value].copy() if rc >= 0 else None)


def host_delaunay(pts, threads=1):
    """exact Delaunay of integer points (host code of the product; no GPU needed)"""
    pts = np.asarray(pts).reshape(-1, 2)
    x = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    cap = 2 * len(x) + 16
    tris = np.zeros((cap, 3), dtype=np.int32)
    k = lib().vsm_host_delaunay(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x),
                                tris.ctypes.data_as(C.c_void_p), cap, threads)
    return tris[:k]


def host_delaunay_split(pts, max_task_points, device_top_points=0):
    """the same triangulation through prepare / independent sub-trees / the merge nodes listed for the
    sub-tree solver's side (at most device_top_points points) / remaining merges (host code only)"""
    pts = np.asarray(pts).reshape(-1, 2)
    x = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    cap = 2 * len(x) + 16
    tris = np.zeros((cap, 3), dtype=np.int32)
    k = lib().vsm_host_delaunay_split(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x),
                                      tris.ctypes.data_as(C.c_void_p), cap, max_task_points, device_top_points)
    return tris[:k]


def delaunay_gpu_split(pts, max_task_points, device_top_points=0, device_kd=False):
    """test hook: sub-trees of the exact Delaunay and the merge nodes of at most device_top_points points on
    the GPU (device_kd: the kd order of the keys too), preparation and the remaining merges on the host"""
    pts = np.asarray(pts).reshape(-1, 2)
    x = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    cap = 2 * len(x) + 16
    tris = np.zeros((cap, 3), dtype=np.int32)
    k = lib().vsm_debug_delaunay_gpu(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x),
                                     tris.ctypes.data_as(C.c_void_p), cap, max_task_points, device_top_points, int(device_kd))
    if k < 0:
        raise VisoMatchError("vsm_debug_delaunay_gpu: HIP error")
    return tris[:k]


def ties(pts, gpu=False):
    """which matches at shared pixels stand for their points: sorted (carried index, right index) pairs where they
    differ, from the host emulation of Triangle's vertex sort or from the GPU's; (pairs, kernel microseconds)"""
    pts = np.asarray(pts).reshape(-1, 2)
    x = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    y = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    cap = 4096
    out = np.zeros((cap, 2), dtype=np.int32)
    us = C.c_double(0)
    if gpu:
        k = lib().vsm_debug_ties_gpu(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), cap,
                                     C.byref(us))
    else:
        k = lib().vsm_host_ties(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), cap)
    if k < 0:
        return None, us.value
    o = out[:k]
    return o[np.lexsort(o.T[::-1])], us.value


def remove_outliers(matches, method, w, h, gpu=False, gpu_ties=False, copies=1, threads=1, **params):
    """Matcher::removeOutliers + computePriorStatistics on a match list: host code of the per-frame path (gpu=False; threads > 1:
    split over fork-join threads the way vsm_match runs a frame's final list) or the GPU-resident chain of the look-ahead path;
    returns (survivors, ranges in device layout, kernel microseconds)"""
    p = default_params()
    for k, v in params.items():
        setattr(p, k, v)
    m = np.ascontiguousarray(matches, dtype=P_MATCH)
    n = len(m)
    out = np.zeros(max(n, 1), dtype=P_MATCH)
    ub, vb = -(-w // p.match_binsize), -(-h // p.match_binsize)
    rg = np.zeros((ub * vb, 16), dtype=np.float32)
    us = C.c_double(0)
    if gpu:
        k = lib().vsm_debug_dc2(C.byref(p), m.ctypes.data_as(C.c_void_p), n, method, int(gpu_ties), copies, out.ctypes.data_as(C.c_void_p),
                                len(out), rg.ctypes.data_as(C.c_void_p), w, h, C.cast(C.byref(us), C.c_void_p))
    elif threads > 1:
        k = lib().vsm_host_outliers_and_prior_threads(C.byref(p), m.ctypes.data_as(C.c_void_p), n, method, out.ctypes.data_as(C.c_void_p), len(out),
                                                      rg.ctypes.data_as(C.c_void_p), w, h, threads)
    else:
        k = lib().vsm_host_outliers_and_prior(C.byref(p), m.ctypes.data_as(C.c_void_p), n, method, out.ctypes.data_as(C.c_void_p), len(out),
                                              rg.ctypes.data_as(C.c_void_p), w, h)
    if k < 0:
        raise VisoMatchError(f"remove_outliers: code {k}")
    return out[:k].copy(), rg, us.value


def local_cpus():
    """the CPUs of the GPU's NUMA node the library keeps its own host threads on (after the first Matcher exists); [] = no
    restriction.  os.sched_setaffinity(0, local_cpus()) puts the calling thread there too."""
    buf = (C.c_int32 * 1024)()
    n = lib().vsm_local_cpus(C.cast(buf, C.c_void_p), 1024)
    return [int(buf[i]) for i in range(min(n, 1024))]


def forkjoin_cpus():
    """the CPUs of the L3 domain the per-frame path's fork-join threads share (after the first Matcher exists); [] = none.
    os.sched_setaffinity(0, forkjoin_cpus()) puts the thread that calls match_features / process beside them."""
    buf = (C.c_int32 * 1024)()
    n = lib().vsm_forkjoin_cpus(C.cast(buf, C.c_void_p), 1024)
    return [int(buf[i]) for i in range(min(n, 1024))]


def default_params():
    p = VsmParams()
    lib().vsm_default_params(C.byref(p))
    return p


def _order_behind_torch(handle, *tensors):
    """device-resident inputs: the library reads them asynchronously on its own stream, so its stream is made to wait for
    whatever torch's current stream holds (the kernels or copies that produce the tensors) - vsm_wait_for_stream"""
    import torch
    dev = tensors[0].device
    rc = lib().vsm_wait_for_stream(handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise VisoMatchError(f"vsm_wait_for_stream failed with {rc}")


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Matcher:
    """Drop-in mirror of the reference Matcher.  Images: (H,W) uint8 numpy arrays (host path,
    vsm_push_back) or CUDA/HIP torch tensors (device-resident path, vsm_push_back_device)."""

    OK, EDIMS, ENOTREADY, EHIP, EARG = 0, -1, -2, -3, -4

    _inputs = None

    def __init__(self, stage_capture=False, options=None, **params):
        """params: fields of the reference's Matcher::parameters; options: measurement / test switches of the handle
        (vsm_set_option: "seq_chunk", "dc_gpu", ...), applied right after creation"""
        L = lib()
        p = default_params()
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown matcher parameter {k}")
            setattr(p, k, v)
        self.params = p
        self.h = L.vsm_create(C.byref(p))
        if not self.h:
            raise VisoMatchError("vsm_create failed: no usable HIP device (the matcher has no CPU path)")
        self.h = C.c_void_p(self.h)
        if stage_capture:
            L.vsm_set_stage_capture(self.h, 1)
        # (tools: VSM_PY_OPTIONS="seq_fstreams=2,seq_chunk=50" reaches handles created by scripts that pass no options)
        env_opts = dict(kv.split("=") for kv in os.environ.get("VSM_PY_OPTIONS", "").split(",") if "=" in kv)
        for k, v in {**env_opts, **(options or {})}.items():
            self.set_option(k, int(v))

    # --- reference API -------------------------------------------------------------------
    def set_intrinsics(self, f, cu, cv, base):
        lib().vsm_set_intrinsics(self.h, f, cu, cv, base)

    def push_back(self, I1, I2=None, replace=False):
        L = lib()
        if _is_torch(I1):
            assert I1.is_cuda and I1.dtype.__str__() == "torch.uint8" and I1.dim() == 2
            h, w = I1.shape
            bpl = I1.stride(0)
            assert I1.stride(1) == 1
            p2 = None
            if I2 is not None:
                assert I2.is_cuda and I2.shape == I1.shape and I2.stride() == I1.stride()
                p2 = C.c_void_p(I2.data_ptr())
            _order_behind_torch(self.h, I1)
            self._inputs = (I1, I2)      # the push reads them asynchronously: kept alive until the next push
            return L.vsm_push_back_device(self.h, C.c_void_p(I1.data_ptr()), p2, w, h, bpl, int(replace))
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        h, w = I1.shape
        p2 = None
        if I2 is not None:
            I2 = np.ascontiguousarray(I2, dtype=np.uint8)
            assert I2.shape == I1.shape
            p2 = I2.ctypes.data_as(C.c_void_p)
        return L.vsm_push_back(self.h, I1.ctypes.data_as(C.c_void_p), p2, w, h, w, int(replace))

    def match_features(self, method, Tr_delta=None):
        tp = None
        if Tr_delta is not None:
            t = np.ascontiguousarray(np.asarray(Tr_delta, dtype=np.float64).reshape(-1)[:12])
            tp = t.ctypes.data_as(C.c_void_p)
        rc = lib().vsm_match(self.h, method, tp)
        if rc not in (self.OK, self.ENOTREADY):
            raise VisoMatchError(f"vsm_match failed with {rc}")
        return rc

    def get_matches(self):
        n = lib().vsm_num_matches(self.h)
        out = np.zeros(n, dtype=P_MATCH)
        if n:
            lib().vsm_get_matches(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def bucket_features(self, max_features, bucket_width, bucket_height):
        lib().vsm_bucket(self.h, max_features, bucket_width, bucket_height)

    def get_gain(self, inliers):
        a = np.ascontiguousarray(inliers, dtype=np.int32)
        return float(lib().vsm_gain(self.h, a.ctypes.data_as(C.c_void_p), len(a)))

    # --- the face the golden drivers use (same names as oracle.bindings.CpuMatcher) ---------
    def match(self, method, Tr=None):
        return self.match_features(method, Tr) == self.OK

    matches = get_matches
    bucket = bucket_features
    gain = get_gain

    def stage(self, s):
        n = lib().vsm_stage_size(self.h, s)
        out = np.zeros(n, dtype=P_MATCH)
        if n:
            lib().vsm_stage_get(self.h, s, out.ctypes.data_as(C.c_void_p), n)
        return out

    def ranges(self):
        n = lib().vsm_num_ranges(self.h)
        out = np.zeros((n, 4, 4), dtype=np.float32)
        if n:
            lib().vsm_get_ranges(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def features(self, which):
        w = FEATURE_SETS[which] if isinstance(which, str) else which
        n = lib().vsm_num_features(self.h, w)
        out = np.zeros((n, 12), dtype=np.int32)
        if n:
            got = lib().vsm_get_features(self.h, w, out.ctypes.data_as(C.c_void_p), n)
            assert got == n
        return out

    def gradients(self, which, full):
        n = lib().vsm_get_gradients(self.h, which, int(full), None, None)
        if n == 0:
            return None, None
        du = np.zeros(n, dtype=np.uint8)
        dv = np.zeros(n, dtype=np.uint8)
        lib().vsm_get_gradients(self.h, which, int(full), du.ctypes.data_as(C.c_void_p), dv.ctypes.data_as(C.c_void_p))
        return du, dv

    def filter_responses(self):
        n = lib().vsm_get_filter_responses(self.h, None, None)
        if n == 0:
            return None, None
        f1 = np.zeros(n, dtype=np.int16)
        f2 = np.zeros(n, dtype=np.int16)
        lib().vsm_get_filter_responses(self.h, f1.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p))
        return f1, f2

    def counters(self):
        c = np.zeros(5, dtype=np.int64)
        lib().vsm_get_counters(self.h, c.ctypes.data_as(C.c_void_p))
        return c

    def timings(self):
        t = np.zeros(5, dtype=np.float64)
        lib().vsm_get_timings(self.h, t.ctypes.data_as(C.c_void_p))
        return dict(zip(("pass1_gpu_us", "pass1_host_us", "pass2_gpu_us", "final_host_us", "total_us"), t.tolist()))

    # --- look-ahead API -------------------------------------------------------------------
    def run_sequence(self, left, right, method, Tr_delta=None, Tr_valid=None, fetch=True):
        """left/right: [F,H,W] uint8 numpy arrays (host) or CUDA torch tensors (resident in HBM);
        returns the list of per-frame match arrays (getMatches() after each frame), or, with
        fetch=False, nothing (the lists stay in the handle: sequence_matches(f))."""
        L = lib()
        if _is_torch(left):
            assert left.is_cuda and left.dim() == 3 and left.stride(2) == 1
            F, h, w = left.shape
            bpl, fs = left.stride(1), left.stride(0)
            _order_behind_torch(self.h, left)
            pl = C.c_void_p(left.data_ptr())
            pr = None
            if right is not None:
                assert right.is_cuda and right.shape == left.shape and right.stride() == left.stride()
                pr = C.c_void_p(right.data_ptr())
            dev = 1
        else:
            left = np.ascontiguousarray(left, dtype=np.uint8)
            F, h, w = left.shape
            bpl, fs = w, w * h
            pl = left.ctypes.data_as(C.c_void_p)
            pr = None
            if right is not None:
                right = np.ascontiguousarray(right, dtype=np.uint8)
                assert right.shape == left.shape
                pr = right.ctypes.data_as(C.c_void_p)
            dev = 0
        tp = vp_ = None
        if Tr_delta is not None:
            t = np.ascontiguousarray(np.asarray(Tr_delta, dtype=np.float64).reshape(F, -1)[:, :12])
            tp = t.ctypes.data_as(C.c_void_p)
            if Tr_valid is not None:
                v = np.ascontiguousarray(np.asarray(Tr_valid).astype(np.uint8))
                vp_ = v.ctypes.data_as(C.c_void_p)
        rc = L.vsm_sequence_run(self.h, pl, pr, fs, dev, F, w, h, bpl, method, tp, vp_)
        if rc != self.OK:
            raise VisoMatchError(f"vsm_sequence_run failed with {rc}")
        return [self.sequence_matches(f) for f in range(F)] if fetch else None

    def sequence_matches(self, f):
        L = lib()
        n = L.vsm_sequence_num_matches(self.h, f)
        a = np.zeros(n, dtype=P_MATCH)
        if n:
            L.vsm_sequence_get_matches(self.h, f, a.ctypes.data_as(C.c_void_p), n)
        return a

    def sequence_path(self):
        """2: the last run_sequence went through the GPU-resident form, 1: through the host-shared form"""
        return lib().vsm_sequence_path(self.h)

    def set_option(self, name, value):
        """measurement / test switch of this handle (vsm_set_option): "seq_serial", "seq_chunk", "seq_v2", ..."""
        if lib().vsm_set_option(self.h, name.encode(), int(value)) != self.OK:
            raise VisoMatchError(f"unknown option {name}")

    def sequence_timings(self):
        t = np.zeros(4, dtype=np.float64)
        lib().vsm_sequence_get_timings(self.h, t.ctypes.data_as(C.c_void_p))
        return dict(zip(("gpu_us", "host_us", "total_us", "chunk"), t.tolist()))

    def set_profiling(self, on, only=None, print_spans=False):
        """HIP-event spans around the library's kernels: all of them, or (only="k_match<16>:pass2") one kernel's launches
        alone - the spans' own event records are packets on every stream and slow the pipeline they measure"""
        L = lib()
        if on and only is not None:
            ids = [i for i in range(L.vsm_num_kernels()) if L.vsm_kernel_name(i).decode() == only]
            if not ids:
                raise VisoMatchError(f"unknown kernel {only}")
            L.vsm_set_profiling(self.h, (1100 if print_spans else 100) + ids[0])
        else:
            L.vsm_set_profiling(self.h, int(bool(on)))

    def kernel_stats(self):
        """{kernel name: (total device ms, launches)} since set_profiling(True)"""
        L = lib()
        n = L.vsm_num_kernels()
        ms = np.zeros(n, dtype=np.float64)
        cnt = np.zeros(n, dtype=np.int64)
        L.vsm_get_kernel_stats(self.h, ms.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
        return {L.vsm_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def close(self):
        if getattr(self, "h", None):
            lib().vsm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiVisualOdometryStereo:
    """K independent stereo sequences in lock-step (vsm_multi_*): process(left, right) is VisualOdometryStereo::process
    for the next pair of every sequence, one launch per kernel over all of them; per sequence the results equal the
    reference's for that sequence alone."""

    def __init__(self, n_sequences, f, cu, cv, base, bucket=(2, 50.0, 50.0), ransac_iters=200, inlier_threshold=2.0,
                 reweighting=True, **match):
        self.params = vo_stereo_params(f, cu, cv, base, bucket, ransac_iters, inlier_threshold, reweighting, **match)
        self.K = int(n_sequences)
        h = lib().vsm_multi_create(C.byref(self.params), self.K)
        if not h:
            raise VisoMatchError("vsm_multi_create failed: no usable HIP device (there is no CPU path)")
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            lib().vsm_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def process(self, left, right):
        """left / right: [K,H,W] uint8, numpy (host) or CUDA torch tensors -> K success flags"""
        L = lib()
        ok = np.zeros(self.K, dtype=np.int32)
        # (checked BEFORE the call: the library reads K images per side from these pointers)
        if _is_torch(left):
            import torch
            if not (_is_torch(right) and left.is_cuda and right.is_cuda and left.dtype == torch.uint8 and right.dtype == torch.uint8 and left.dim() == 3
                    and left.shape == right.shape and left.shape[0] == self.K and left.stride() == right.stride() and left.stride(2) == 1):
                raise VisoMatchError(f"process: left / right must be uint8 CUDA tensors [K={self.K},H,W] of equal shape and strides, unit stride along W")
            K, h, w = left.shape
            torch.cuda.current_stream(left.device).synchronize()   # (the library reads the images on its own stream)
            rc = L.vsm_multi_process(self.h, C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr()), left.stride(0), 1, w, h, left.stride(1),
                                     ok.ctypes.data_as(C.c_void_p))
        else:
            left = np.ascontiguousarray(left, dtype=np.uint8)
            right = np.ascontiguousarray(right, dtype=np.uint8)
            if left.ndim != 3 or left.shape != right.shape or left.shape[0] != self.K:
                raise VisoMatchError(f"process: left / right must be uint8 arrays [K={self.K},H,W] of equal shape, got {left.shape} / {right.shape}")
            K, h, w = left.shape
            rc = L.vsm_multi_process(self.h, left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), w * h, 0, w, h, w,
                                     ok.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise VisoMatchError(f"vsm_multi_process failed with {rc}")
        return ok.astype(bool)

    def get_motion(self, k):
        t = np.zeros(16)
        lib().vsm_multi_get_motion(self.h, k, t.ctypes.data_as(C.c_void_p))
        return t.reshape(4, 4)

    def motion_valid(self, k):
        return bool(lib().vsm_multi_motion_valid(self.h, k))

    def get_matches(self, k, bucketed=True):
        n = lib().vsm_multi_num_matches(self.h, k, int(bucketed))
        out = np.zeros(n, dtype=P_MATCH)
        if n:
            lib().vsm_multi_get_matches(self.h, k, int(bucketed), out.ctypes.data_as(C.c_void_p), n)
        return out

    def get_inlier_indices(self, k):
        n = lib().vsm_multi_num_inliers(self.h, k)
        out = np.zeros(n, dtype=np.int32)
        if n:
            lib().vsm_multi_get_inliers(self.h, k, out.ctypes.data_as(C.c_void_p), n)
        return out

    def timings(self):
        t = np.zeros(4, dtype=np.float64)
        lib().vsm_multi_get_timings(self.h, t.ctypes.data_as(C.c_void_p))
        return dict(zip(("features_us", "pass1_us", "pass2_us", "egomotion_us"), t.tolist()))


class VisualOdometryStereo:
    """Mirror of the reference's VisualOdometryStereo (viso/viso_stereo.h:28-88): process() runs
    pushBack + matchFeatures(2, Tr_delta) + bucketFeatures + egomotion and keeps Tr_delta."""

    def __init__(self, f, cu, cv, base, bucket=(2, 50.0, 50.0), ransac_iters=200, inlier_threshold=2.0,
                 reweighting=True, **match):
        self.params = vo_stereo_params(f, cu, cv, base, bucket, ransac_iters, inlier_threshold, reweighting, **match)
        h = lib().vsm_vo_stereo_create(C.byref(self.params))
        if not h:
            raise VisoMatchError("vsm_vo_stereo_create failed: no usable HIP device (there is no CPU path)")
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            lib().vsm_vo_stereo_destroy(self.h)
            self.h = None

    __del__ = close

    def get_motion(self):
        t = np.zeros(16)
        lib().vsm_vo_stereo_get_motion(self.h, t.ctypes.data_as(C.c_void_p))
        return t.reshape(4, 4)

    def process(self, I1, I2, replace=False):
        """-> (success, Tr_valid before the call, Tr_delta before, Tr_delta after)"""
        L = lib()
        valid = bool(L.vsm_vo_stereo_motion_valid(self.h))
        tin = self.get_motion()
        if _is_torch(I1):
            assert I1.is_cuda and I2.is_cuda and I1.shape == I2.shape and I1.stride() == I2.stride()
            h, w = I1.shape
            _order_behind_torch(L.vsm_vo_stereo_matcher(self.h), I1)
            ok = L.vsm_vo_stereo_process_device(self.h, C.c_void_p(I1.data_ptr()), C.c_void_p(I2.data_ptr()), w, h,
                                                I1.stride(0), int(replace))
        else:
            I1 = np.ascontiguousarray(I1, dtype=np.uint8)
            I2 = np.ascontiguousarray(I2, dtype=np.uint8)
            h, w = I1.shape
            ok = L.vsm_vo_stereo_process(self.h, I1.ctypes.data_as(C.c_void_p), I2.ctypes.data_as(C.c_void_p), w, h, w,
                                         int(replace))
        return bool(ok), valid, tin, self.get_motion()

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=P_MATCH)
        ok = lib().vsm_vo_stereo_process_matches(self.h, m.ctypes.data_as(C.c_void_p), len(m))
        return bool(ok), self.get_motion()

    def get_matches(self):
        n = lib().vsm_vo_stereo_num_matches(self.h)
        out = np.zeros(n, dtype=P_MATCH)
        if n:
            lib().vsm_vo_stereo_get_matches(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def get_inlier_indices(self):
        n = lib().vsm_vo_stereo_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            lib().vsm_vo_stereo_get_inliers(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def get_number_of_matches(self):
        return lib().vsm_vo_stereo_num_matches(self.h)

    def get_number_of_inliers(self):
        return lib().vsm_vo_stereo_num_inliers(self.h)

    def get_gain(self, inliers):
        a = np.ascontiguousarray(inliers, dtype=np.int32)
        return float(lib().vsm_vo_stereo_gain(self.h, a.ctypes.data_as(C.c_void_p), len(a)))

    def timings(self):
        t = np.zeros(4)
        lib().vsm_vo_stereo_get_timings(self.h, t.ctypes.data_as(C.c_void_p))
        return t

    # the face the golden drivers use (same names as oracle.bindings.OracleStereoVO)
    bucketed = get_matches
    inliers = get_inlier_indices


class VisualOdometryMono:
    """Mirror of the reference's VisualOdometryMono (viso/viso_mono.h:28-90): process() runs
    pushBack + matchFeatures(0) + bucketFeatures + the monocular egomotion."""

    def __init__(self, f, cu, cv, bucket=(2, 50.0, 50.0), **kw):
        self.params = vo_mono_params(f, cu, cv, bucket, **kw)
        h = lib().vsm_vo_mono_create(C.byref(self.params))
        if not h:
            raise VisoMatchError("vsm_vo_mono_create failed: no usable HIP device (there is no CPU path)")
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            lib().vsm_vo_mono_destroy(self.h)
            self.h = None

    __del__ = close

    def get_motion(self):
        t = np.zeros(16)
        lib().vsm_vo_mono_get_motion(self.h, t.ctypes.data_as(C.c_void_p))
        return t.reshape(4, 4)

    def process(self, I, replace=False):
        """-> (success, Tr_delta after the call)"""
        L = lib()
        if _is_torch(I):
            assert I.is_cuda and I.dim() == 2 and I.stride(1) == 1
            h, w = I.shape
            _order_behind_torch(L.vsm_vo_mono_matcher(self.h), I)
            ok = L.vsm_vo_mono_process_device(self.h, C.c_void_p(I.data_ptr()), w, h, I.stride(0), int(replace))
        else:
            I = np.ascontiguousarray(I, dtype=np.uint8)
            h, w = I.shape
            ok = L.vsm_vo_mono_process(self.h, I.ctypes.data_as(C.c_void_p), w, h, w, int(replace))
        return bool(ok), self.get_motion()

    def process_matches(self, m):
        m = np.ascontiguousarray(m, dtype=P_MATCH)
        ok = lib().vsm_vo_mono_process_matches(self.h, m.ctypes.data_as(C.c_void_p), len(m))
        return bool(ok), self.get_motion()

    def get_matches(self):
        n = lib().vsm_vo_mono_num_matches(self.h)
        out = np.zeros(n, dtype=P_MATCH)
        if n:
            lib().vsm_vo_mono_get_matches(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def get_inlier_indices(self):
        n = lib().vsm_vo_mono_num_inliers(self.h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            lib().vsm_vo_mono_get_inliers(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def get_number_of_matches(self):
        return lib().vsm_vo_mono_num_matches(self.h)

    def get_number_of_inliers(self):
        return lib().vsm_vo_mono_num_inliers(self.h)

    def get_gain(self, inliers):
        a = np.ascontiguousarray(inliers, dtype=np.int32)
        return float(lib().vsm_vo_mono_gain(self.h, a.ctypes.data_as(C.c_void_p), len(a)))

    def timings(self):
        t = np.zeros(10)
        lib().vsm_vo_mono_get_timings(self.h, t.ctypes.data_as(C.c_void_p))
        return t

    def device_svd(self):
        return bool(lib().vsm_vo_mono_device_svd(self.h))

    bucketed = get_matches
    inliers = get_inlier_indices
