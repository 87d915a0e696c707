"""ctypes loader for the CPU ORACLE (oracle/libsf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from multi_robot_slam_separators_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsf_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/libsf_oracle.so with gcc (Makefile in this directory)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in
                                          ("sf_oracle.c", "sf_oracle_pnp.c", "sf_oracle_ba.c", "sf_oracle_extract.c", "sf_oracle_gftt.c", "sf_oracle_lk.c", "sf_oracle.h",
                                           "sf_oracle_internal.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        # size the OpenMP team to the CPU share of this process (cgroup quota), not to the host's CPU
        # count: an oversubscribed quota is throttled by the kernel and the timing means nothing
        from multi_robot_slam_separators_amd.hostinfo import cpu_share
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_share()))
        L = C.CDLL(_LIB_PATH)
        L.sfo_set_num_threads.restype = None
        L.sfo_set_num_threads.argtypes = [C.c_int]
        L.sfo_set_num_threads(cpu_share())     # explicit: another OpenMP user in the process may have shrunk the team
        P = C.POINTER
        L.sfo_find_matches.restype = C.c_int
        L.sfo_find_matches.argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
            C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
            C.c_double, C.c_int, C.c_void_p, C.c_int, P(C.c_int), C.c_void_p, C.c_void_p]
        L.sfo_match_global.restype = C.c_int
        L.sfo_match_global.argtypes = [
            C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int), C.c_int]
        L.sfo_match_guided.restype = C.c_int
        L.sfo_match_guided.argtypes = [
            P(_abi.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)]
        L.sfo_estimate_motion_3d3d.restype = C.c_int
        L.sfo_estimate_motion_3d3d.argtypes = [
            P(_abi.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
            P(Motion), C.c_void_p]
        L.sfo_estimate_transform.restype = C.c_int
        L.sfo_estimate_transform.argtypes = [P(_abi.Params), P(_abi.Features), P(_abi.Features),
                                             P(_abi.Result)]
        L.sfo_estimate_transform_dbg.restype = C.c_int
        L.sfo_estimate_transform_dbg.argtypes = [
            P(_abi.Params), P(_abi.Features), P(_abi.Features), P(_abi.Result),
            C.c_void_p, C.c_void_p, P(C.c_int), C.c_void_p, C.c_void_p, P(C.c_int)]
        L.sfo_estimate_transform_batch.restype = C.c_int
        L.sfo_estimate_transform_batch.argtypes = [
            P(_abi.Params), P(_abi.Features), P(_abi.Features), C.c_int, C.c_void_p, C.c_int]
        L.sfo_fit_rigid.restype = None
        L.sfo_fit_rigid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sfo_canon_log.restype = C.c_double
        L.sfo_canon_log.argtypes = [C.c_double]
        L.sfo_sample_triplet.restype = None
        L.sfo_sample_triplet.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.sfo_num_threads.restype = C.c_int
        L.sfo_detect_corners.restype = C.c_int
        L.sfo_detect_corners.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                         C.c_void_p, C.c_int32, P(C.c_int32), C.c_void_p]
        L.sfo_stereo_correspondences.restype = C.c_int
        L.sfo_stereo_correspondences.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                 P(_abi.StereoFlowParams), C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int32)]
        L.sfo_pyr_down.restype = None
        L.sfo_pyr_down.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.sfo_scharr_deriv.restype = None
        L.sfo_scharr_deriv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.sfo_extract_keyframe.restype = C.c_int
        L.sfo_extract_keyframe.argtypes = [
            C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
            P(_abi.StereoCamera), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int32)]
        L.sfo_estimate_motion_3d2d.restype = C.c_int
        L.sfo_estimate_motion_3d2d.argtypes = [
            P(_abi.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
            P(Motion), C.c_void_p]
        L.sfo_sample_quad.restype = None
        L.sfo_sample_quad.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.sfo_quartic_roots.restype = C.c_int
        L.sfo_quartic_roots.argtypes = [C.c_void_p, C.c_void_p]
        L.sfo_p3p.restype = C.c_int
        L.sfo_p3p.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfo_canon_atan2.restype = C.c_double
        L.sfo_canon_atan2.argtypes = [C.c_double, C.c_double]
        _lib = L
    return _lib


class Motion(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 12),
        ("is_null", C.c_int),
        ("variance", C.c_double),
        ("variance_ang", C.c_double),
        ("matches", C.c_int),
        ("inliers", C.c_int),
        ("ransac_best_iteration", C.c_int),
        ("ransac_iterations_run", C.c_int),
        ("ransac_best_count", C.c_int),
        ("refine_rounds", C.c_int),
    ]


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


def find_matches(local, received, local_used=(), other_used=(), ignored_pairs=(),
                 netvlad_distance=0.13, max_matches_nb=20):
    """DataHandler.find_matches (data_handler.py:166-209).  Returns (matches, row_min, row_arg)
    with matches a structured array (idx_local, idx_other, distance)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    received = np.ascontiguousarray(received, dtype=np.float64)
    n_l, dim = local.shape
    n_r = received.shape[0]
    lu = np.ascontiguousarray(local_used, dtype=np.int32)
    ou = np.ascontiguousarray(other_used, dtype=np.int32)
    ig = np.ascontiguousarray(ignored_pairs, dtype=np.int32).reshape(-1, 2)
    cap = max(1, min(n_l, max_matches_nb))
    out = np.zeros(cap, dtype=_abi.MATCH_DTYPE)
    n = C.c_int(0)
    row_min = np.zeros(n_l, dtype=np.float64)
    row_arg = np.zeros(n_l, dtype=np.int32)
    rc = lib().sfo_find_matches(_ptr(local), n_l, _ptr(received), n_r, dim, _ptr(lu), lu.size,
                                _ptr(ou), ou.size, _ptr(ig), ig.shape[0], netvlad_distance,
                                max_matches_nb, out.ctypes.data, cap, C.byref(n),
                                row_min.ctypes.data, row_arg.ctypes.data)
    if rc != 0:
        raise RuntimeError("sfo_find_matches -> %s" % _abi.STATUS_NAMES.get(rc, rc))
    return out[: n.value], row_min, row_arg


def match_global(desc_from, desc_to, nndr=0.6, has3d_from=True, has3d_to=True, desc_type=0):
    """desc_type 1: rows of float32 handed over as their bytes (rows x 4 * dims uint8) or as a float32 array."""
    if desc_type == 1:
        desc_from = np.ascontiguousarray(desc_from, dtype=np.float32).view(np.uint8) if np.asarray(desc_from).dtype != np.uint8 else desc_from
        desc_to = np.ascontiguousarray(desc_to, dtype=np.float32).view(np.uint8) if np.asarray(desc_to).dtype != np.uint8 else desc_to
    df = np.ascontiguousarray(desc_from, dtype=np.uint8)
    dt = np.ascontiguousarray(desc_to, dtype=np.uint8)
    cap = max(1, df.shape[0], dt.shape[0])
    cf = np.zeros(cap, dtype=np.uint16)
    ct = np.zeros(cap, dtype=np.uint16)
    n, wf, wt, wt2 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    cols = df.shape[1] if df.ndim == 2 and df.shape[0] else (dt.shape[1] if dt.ndim == 2 else 0)
    rc = lib().sfo_match_global(_ptr(df), df.shape[0], _ptr(dt), dt.shape[0], cols,
                                C.c_float(nndr), int(has3d_from), int(has3d_to), cf.ctypes.data,
                                ct.ctypes.data, C.byref(n), C.byref(wf), C.byref(wt), C.byref(wt2), int(desc_type))
    if rc != 0:
        raise RuntimeError("sfo_match_global -> %d" % rc)
    return cf[: n.value].copy(), ct[: n.value].copy(), wf.value, wt.value, wt2.value


def match_guided(params, guess, f_from, f_to):
    guess = np.ascontiguousarray(guess, dtype=np.float32).reshape(12)
    cap = max(1, f_from.desc.shape[0], f_to.desc.shape[0])
    cf = np.zeros(cap, dtype=np.uint16)
    ct = np.zeros(cap, dtype=np.uint16)
    n, wf, wt, wt2, ao = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().sfo_match_guided(C.byref(params), guess.ctypes.data, _ptr(f_from.desc),
                                _ptr(f_from.xyz), _ptr(f_from.kpts), f_from.desc.shape[0],
                                _ptr(f_to.desc), _ptr(f_to.kpts), f_to.desc.shape[0],
                                int(f_to.xyz.shape[0] > 0), f_from.desc.shape[1], cf.ctypes.data,
                                ct.ctypes.data, C.byref(n), C.byref(wf), C.byref(wt), C.byref(wt2),
                                C.byref(ao))
    if rc != 0:
        raise RuntimeError("sfo_match_guided -> %d" % rc)
    return cf[: n.value].copy(), ct[: n.value].copy(), wf.value, wt.value, wt2.value, ao.value


def estimate_motion_3d3d(params, xyz_from, xyz_to, corr_from, corr_to):
    xf = np.ascontiguousarray(xyz_from, dtype=np.float32)
    xt = np.ascontiguousarray(xyz_to, dtype=np.float32)
    cf = np.ascontiguousarray(corr_from, dtype=np.uint16)
    ct = np.ascontiguousarray(corr_to, dtype=np.uint16)
    mo = Motion()
    mask = np.zeros(max(1, cf.size), dtype=np.uint8)
    rc = lib().sfo_estimate_motion_3d3d(C.byref(params), _ptr(xf), _ptr(xt), _ptr(cf), _ptr(ct),
                                        cf.size, C.byref(mo), mask.ctypes.data)
    if rc != 0:
        raise RuntimeError("sfo_estimate_motion_3d3d -> %d" % rc)
    return mo, mask[: cf.size]


def estimate_motion_3d2d(params, xyz_from, kp_to, xyz_to, corr_from, corr_to):
    """util3d::estimateMotion3DTo2D restatement (sf_oracle_pnp.c).  xyz_to may be None."""
    xf = np.ascontiguousarray(xyz_from, dtype=np.float32)
    kt = np.ascontiguousarray(kp_to, dtype=_abi.KEYPOINT_DTYPE)
    xt = None if xyz_to is None else np.ascontiguousarray(xyz_to, dtype=np.float32)
    cf = np.ascontiguousarray(corr_from, dtype=np.uint16)
    ct = np.ascontiguousarray(corr_to, dtype=np.uint16)
    mo = Motion()
    mask = np.zeros(max(1, cf.size), dtype=np.uint8)
    rc = lib().sfo_estimate_motion_3d2d(C.byref(params), _ptr(xf), kt.ctypes.data, _ptr(xt), _ptr(cf), _ptr(ct),
                                        cf.size, C.byref(mo), mask.ctypes.data)
    if rc != 0:
        raise RuntimeError("sfo_estimate_motion_3d2d -> %d" % rc)
    return mo, mask[: cf.size]


def sample_quad(seed, it, attempt, m):
    out = np.zeros(4, dtype=np.uint32)
    lib().sfo_sample_quad(seed, it, attempt, m, out.ctypes.data)
    return out


def quartic_roots(c):
    c = np.ascontiguousarray(c, dtype=np.float64)
    r = np.zeros(4)
    n = lib().sfo_quartic_roots(c.ctypes.data, r.ctypes.data)
    return r[:n]


def p3p(P, f):
    P = np.ascontiguousarray(P, dtype=np.float64)
    f = np.ascontiguousarray(f, dtype=np.float64)
    R = np.zeros((4, 9))
    t = np.zeros((4, 3))
    n = lib().sfo_p3p(P.ctypes.data, f.ctypes.data, R.ctypes.data, t.ctypes.data)
    return R[:n].reshape(n, 3, 3), t[:n]


def canon_atan2(y, x):
    return lib().sfo_canon_atan2(float(y), float(x))


def estimate_transform(params, f_from, f_to, debug=False):
    """estimateTransformation (stereoCamGeometricTools.cpp:122-178).  Returns a numpy record of
    RESULT_DTYPE (and the two correspondence lists when debug=True)."""
    res = np.zeros(1, dtype=_abi.RESULT_DTYPE)
    a, b = f_from.c_struct(), f_to.c_struct()
    if not debug:
        rc = lib().sfo_estimate_transform(C.byref(params), C.byref(a), C.byref(b),
                                          C.cast(res.ctypes.data, C.POINTER(_abi.Result)))
        if rc != 0:
            raise RuntimeError("sfo_estimate_transform -> %s" % _abi.STATUS_NAMES.get(rc, rc))
        return res[0]
    cap = max(1, f_from.desc.shape[0], f_to.desc.shape[0])
    bufs = [np.zeros(cap, dtype=np.uint16) for _ in range(4)]
    n1, n2 = C.c_int(), C.c_int()
    rc = lib().sfo_estimate_transform_dbg(C.byref(params), C.byref(a), C.byref(b),
                                          C.cast(res.ctypes.data, C.POINTER(_abi.Result)),
                                          bufs[0].ctypes.data, bufs[1].ctypes.data, C.byref(n1),
                                          bufs[2].ctypes.data, bufs[3].ctypes.data, C.byref(n2))
    if rc != 0:
        raise RuntimeError("sfo_estimate_transform_dbg -> %s" % _abi.STATUS_NAMES.get(rc, rc))
    return res[0], (bufs[0][: n1.value], bufs[1][: n1.value]), (bufs[2][: n2.value], bufs[3][: n2.value])


def estimate_transform_batch(params, feats_from, feats_to, threads=1):
    n = len(feats_from)
    fa = _abi.features_array(feats_from)
    ta = _abi.features_array(feats_to)
    res = np.zeros(n, dtype=_abi.RESULT_DTYPE)
    rc = lib().sfo_estimate_transform_batch(C.byref(params), fa, ta, n, res.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("sfo_estimate_transform_batch -> %s" % _abi.STATUS_NAMES.get(rc, rc))
    return res


def fit_rigid(src, dst):
    src = np.ascontiguousarray(src, dtype=np.float64)
    dst = np.ascontiguousarray(dst, dtype=np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    lib().sfo_fit_rigid(src.ctypes.data, dst.ctypes.data, src.shape[0], R.ctypes.data, t.ctypes.data)
    return R.reshape(3, 3), t


def canon_log(x):
    return lib().sfo_canon_log(float(x))


def sample_triplet(seed, it, attempt, m):
    out = np.zeros(3, dtype=np.uint32)
    lib().sfo_sample_triplet(seed, it, attempt, m, out.ctypes.data)
    return out


def num_threads():
    return lib().sfo_num_threads()


def extract_keyframe(image, kpts, right_x, status, cam, tests):
    """sfo_extract_keyframe: image uint8 [h, w] (C-contiguous rows, pitch = stride), kpts KEYPOINT_DTYPE [n],
    right_x float32 [n] or None, status uint8 [n] or None, tests int8 [8 * bytes, 4].
    Returns (desc [rows, bytes] u8, xyz [rows, 3] f32, kpts [rows])."""
    L = lib()
    image = np.asarray(image, np.uint8)
    assert image.ndim == 2 and image.strides[1] == 1
    kpts = np.ascontiguousarray(kpts, dtype=_abi.KEYPOINT_DTYPE)
    n = kpts.shape[0]
    tests = np.ascontiguousarray(tests, np.int8).reshape(-1, 4)
    nbytes = tests.shape[0] // 8
    rx = None if right_x is None else np.ascontiguousarray(right_x, np.float32)
    stt = None if status is None else np.ascontiguousarray(status, np.uint8)
    desc = np.zeros((max(n, 1), nbytes), np.uint8)
    xyz = np.zeros((max(n, 1), 3), np.float32)
    kout = np.zeros(max(n, 1), _abi.KEYPOINT_DTYPE)
    rows = C.c_int32()
    rc = L.sfo_extract_keyframe(image.ctypes.data, image.shape[1], image.shape[0], image.strides[0],
                                kpts.ctypes.data if n else None, None if rx is None else rx.ctypes.data,
                                None if stt is None else stt.ctypes.data, n, C.byref(cam), tests.ctypes.data, nbytes,
                                desc.ctypes.data, xyz.ctypes.data, kout.ctypes.data, C.byref(rows))
    if rc != 0:
        raise RuntimeError("sfo_extract_keyframe failed: %d" % rc)
    r = rows.value
    return desc[:r].copy(), xyz[:r].copy(), kout[:r].copy()


def detect_corners(image, max_corners=1000, quality_level=0.001, min_distance=3.0, want_eig=False):
    """sfo_detect_corners: image uint8 [h, w] (unit column stride).  Returns kpts (KEYPOINT_DTYPE) [, eig float32 [h, w]]."""
    L = lib()
    image = np.asarray(image, np.uint8)
    assert image.ndim == 2 and image.strides[1] == 1
    h, w = image.shape
    cap = w * h
    kp = np.zeros(cap, _abi.KEYPOINT_DTYPE)
    eig = np.zeros((h, w), np.float32) if want_eig else None
    n = C.c_int32()
    rc = L.sfo_detect_corners(image.ctypes.data, w, h, image.strides[0], max_corners, quality_level, min_distance,
                              kp.ctypes.data, cap, C.byref(n), None if eig is None else eig.ctypes.data)
    if rc != 0:
        raise RuntimeError("sfo_detect_corners failed: %d" % rc)
    kp = kp[:min(n.value, cap)].copy()
    return (kp, eig) if want_eig else kp


def pyr_down(image):
    """sfo_pyr_down: cv::pyrDown of an 8-bit image [h, w] (unit column stride)."""
    L = lib()
    image = np.asarray(image, np.uint8)
    assert image.ndim == 2 and image.strides[1] == 1
    h, w = image.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    L.sfo_pyr_down(image.ctypes.data, w, h, image.strides[0], out.ctypes.data)
    return out


def scharr_deriv(image):
    """sfo_scharr_deriv: int16 [h, w, 2] {Ix, Iy} of lkpyramid.cpp's calcSharrDeriv."""
    L = lib()
    image = np.asarray(image, np.uint8)
    assert image.ndim == 2 and image.strides[1] == 1
    h, w = image.shape
    out = np.zeros((h, w, 2), np.int16)
    L.sfo_scharr_deriv(image.ctypes.data, w, h, image.strides[0], out.ctypes.data)
    return out


def stereo_correspondences(left, right, kpts, params=None, want_levels=False):
    """sfo_stereo_correspondences: left / right uint8 [h, w] with the SAME row stride.  Returns right_xy float32 [n, 2],
    status uint8 [n], err float32 [n] [, pyramid levels built]."""
    L = lib()
    left, right = np.asarray(left, np.uint8), np.asarray(right, np.uint8)
    assert left.shape == right.shape and left.strides == right.strides and left.strides[1] == 1
    h, w = left.shape
    kpts = np.ascontiguousarray(kpts, _abi.KEYPOINT_DTYPE)
    n = len(kpts)
    prm = params if params is not None else _abi.stereo_flow_params()
    xy = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    lv = C.c_int32()
    rc = L.sfo_stereo_correspondences(left.ctypes.data, right.ctypes.data, w, h, left.strides[0], kpts.ctypes.data, n,
                                      C.byref(prm), xy.ctypes.data, st.ctypes.data, err.ctypes.data, C.byref(lv))
    if rc != 0:
        raise RuntimeError("sfo_stereo_correspondences failed: %d" % rc)
    return (xy, st, err, lv.value) if want_levels else (xy, st, err)
