"""ctypes binding of the product library libsepfinder.so (C-ABI: include/sepfinder.h).

There is NO CPU fallback: if the shared library is missing, or no GPU is visible, every entry
point raises.  PyTorch is imported first (when installed) only so that this process ends up with
ONE HIP runtime: torch bundles libamdhip64.so (SONAME libamdhip64.so.7) and the library's
NEEDED entry resolves to the copy that is already loaded.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEPFINDER_LIB") or os.path.join(_HERE, "libsepfinder.so")   # override: A/B builds
_lib = None


class SepfinderError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (_abi.STATUS_NAMES.get(code, code), msg))
        self.code = code


def build(force=False):
    """Compile the HIP sources for gfx950 (csrc/Makefile, hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the separator-finder path)" % LIB_PATH)
    try:  # one HIP runtime per process (see module docstring)
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the C-ABI itself
        for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                break
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    sig = {
        "sf_abi_version": (C.c_int, []),
        "sf_default_params": (None, [P(_abi.Params)]),
        "sf_create": (C.c_int, [P(_abi.Params), C.c_int, P(vp)]),
        "sf_destroy": (None, [vp]),
        "sf_last_error": (C.c_char_p, [vp]),
        "sf_get_params": (C.c_int, [vp, P(_abi.Params)]),
        "sf_set_stream": (C.c_int, [vp, vp]),
        "sf_synchronize": (C.c_int, [vp]),
        "sf_nn_append_local": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_append_received": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_append_local_f32_device": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_append_received_f32_device": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_append_local_f16_device": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_append_received_f16_device": (C.c_int, [vp, vp, i32, i32]),
        "sf_nn_sizes": (C.c_int, [vp, P(i32), P(i32)]),
        "sf_nn_mark_local_used": (C.c_int, [vp, i32]),
        "sf_nn_mark_other_used": (C.c_int, [vp, i32]),
        "sf_nn_ignore_pair": (C.c_int, [vp, i32, i32]),
        "sf_nn_reset": (C.c_int, [vp]),
        "sf_nn_set_precision": (C.c_int, [vp, i32]),
        "sf_set_option": (C.c_int, [vp, i32, i32]),
        "sf_nn_find_matches": (C.c_int, [vp, vp, i32, P(i32)]),
        "sf_nn_last_row_minima": (C.c_int, [vp, vp, vp, i32]),
        "sf_nn_last_filter_dims": (C.c_int, [vp, P(i32)]),
        "sf_nn_walk": (C.c_int, [vp, vp, vp, i32, i32, vp, i32, P(i32)]),
        "sf_store_add_keyframe": (C.c_int, [vp, P(_abi.Features), P(i32)]),
        "sf_store_add_keyframes_device": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, P(i32)]),
        "sf_brief_set_pattern": (C.c_int, [vp, vp, i32]),
        "sf_brief_get_pattern": (C.c_int, [vp, vp, i32, P(i32)]),
        "sf_extract_keyframe_device": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, i32, P(_abi.StereoCamera),
                                                 P(i32), P(i32), vp, vp, vp]),
        "sf_detect_corners_device": (C.c_int, [vp, vp, i32, i32, i32, i32, C.c_double, C.c_double, vp, i32, P(i32)]),
        "sf_stereo_flow_defaults": (None, [P(_abi.StereoFlowParams)]),
        "sf_detector_defaults": (None, [P(_abi.DetectorParams)]),
        "sf_get_features_and_descriptor": (C.c_int, [vp, vp, vp, i32, i32, i32, P(_abi.StereoCamera), P(_abi.DetectorParams),
                                                     P(_abi.StereoFlowParams), vp, vp, vp, i32, P(i32), P(i32)]),
        "sf_stereo_correspondences_device": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, i32, P(_abi.StereoFlowParams),
                                                       vp, vp, vp, vp]),
        "sf_netvlad_load": (C.c_int, [vp, P(_abi.NetvladWeights)]),
        "sf_netvlad_infer_device": (C.c_int, [vp, vp, i32, i32, vp, i32]),
        "sf_netvlad_infer_batch_device": (C.c_int, [vp, vp, i32, i32, i32, vp, i32]),
        "sf_store_size": (C.c_int, [vp, P(i32)]),
        "sf_store_clear": (C.c_int, [vp]),
        "sf_estimate_transform": (C.c_int, [vp, P(_abi.Features), P(_abi.Features), vp]),
        "sf_estimate_transform_batch": (C.c_int, [vp, P(_abi.Features), P(_abi.Features), i32, vp]),
        "sf_verify_pairs": (C.c_int, [vp, vp, vp, i32, vp]),
        "sf_verify_pairs_device": (C.c_int, [vp, vp, vp, i32, vp]),
        "sf_verify_matches_device": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "sf_find_matches_and_verify_device": (C.c_int, [vp, i32, i32, vp, i32, C.POINTER(i32), vp]),
        "sf_compact_accepted_device": (C.c_int, [vp, vp, i32, vp, vp, P(i32)]),
        "sf_compact_accepted_device_async": (C.c_int, [vp, vp, i32, vp, vp, vp]),
        "sf_step_issue": (C.c_int, [vp, i32, i32]),
        "sf_step_retire": (C.c_int, [vp, P(_abi.StepResult)]),
        "sf_memcpy_device_async": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
        "sf_step_mirror": (C.c_int, [vp, vp, vp, i32]),
        "sf_step_mirror_pair": (C.c_int, [vp, vp, vp, vp, vp, i32]),
        "sf_step_mirror_streams": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "sf_accept_stream_set": (C.c_int, [vp, i32, vp, vp, vp, i32, vp, vp]),
        "sf_accept_stream_select": (C.c_int, [vp, i32]),
        "sf_accept_stream_status": (C.c_int, [vp, P(i32), P(i32)]),
        "sf_compact_accepted_indexed_mirrored_device_async": (C.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
        "sf_last_match_results": (C.c_int, [vp, P(vp), P(vp), P(i32)]),
        "sf_compact_accepted_indexed_device_async": (C.c_int, [vp, vp, vp, i32, vp, vp, vp]),
        "sf_debug_correspondences": (C.c_int, [vp, i32, i32, vp, vp, i32, P(i32)]),
        "sf_debug_pass_state": (C.c_int, [vp, i32, i32, vp, P(i32), P(i32), P(i32)]),
        "sf_debug_counters": (C.c_int, [vp, vp, i32]),
        "sf_debug_guided_points": (C.c_int, [vp, i32, vp, vp, P(i32)]),
        "sf_debug_plan_workspace": (C.c_int, [P(_abi.Params), i32, i32, i32, i32, i32, P(i64), i32]),
        "sf_pack_separators": (C.c_int, [vp, i32, C.c_int8, C.c_int8, vp, vp, vp, vp, vp]),
        "sf_comm_unique_id": (C.c_int, [vp, i32]),
        "sf_comm_init": (C.c_int, [vp, vp, i32, i32]),
        "sf_comm_destroy": (C.c_int, [vp]),
        "sf_allgather_separators": (C.c_int, [vp, vp, i32, vp, i32, vp]),
        "sf_allgather_separators_device": (C.c_int, [vp, vp, vp, i32]),
        "sf_allgather_bytes_device": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "sf_nn_row_minima_device": (C.c_int, [vp, vp, vp, vp]),
        "sf_nn_walk_device": (C.c_int, [vp, vp, vp, vp, i32, i32, vp, i32, vp]),
        "sf_get_features_and_descriptor_batch_device": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, C.c_size_t, vp, vp, vp,
                                                                   P(i32), vp, vp, vp, vp]),
        "sf_prof_enable": (C.c_int, [vp, C.c_int]),
        "sf_prof_select": (C.c_int, [vp, C.c_uint32]),
        "sf_stream_placement": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
        "sf_streams_prepare": (C.c_int, [vp]),
        "sf_prof_reset": (C.c_int, [vp]),
        "sf_prof_get": (C.c_int, [vp, C.c_int, P(i64), P(C.c_double)]),
        "sf_kernel_name": (C.c_char_p, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    if L.sf_abi_version() != _abi.SF_ABI_VERSION:
        raise ImportError("%s reports ABI version %d, this binding is written for %d (include/sepfinder.h): rebuild the "
                          "library" % (LIB_PATH, L.sf_abi_version(), _abi.SF_ABI_VERSION))
    _lib = L
    return L


EXPORTED = [
    "sf_abi_version", "sf_default_params", "sf_create", "sf_destroy", "sf_last_error", "sf_get_params",
    "sf_set_stream", "sf_synchronize", "sf_nn_append_local", "sf_nn_append_received",
    "sf_nn_append_local_f32_device", "sf_nn_append_received_f32_device", "sf_nn_append_local_f16_device",
    "sf_nn_append_received_f16_device", "sf_nn_sizes",
    "sf_nn_mark_local_used", "sf_nn_mark_other_used", "sf_nn_ignore_pair", "sf_nn_reset",
    "sf_nn_set_precision",
    "sf_set_option",
    "sf_nn_find_matches", "sf_nn_last_row_minima", "sf_nn_last_filter_dims", "sf_nn_walk", "sf_store_add_keyframe",
    "sf_store_add_keyframes_device", "sf_store_size", "sf_store_clear",
    "sf_brief_set_pattern", "sf_brief_get_pattern", "sf_extract_keyframe_device", "sf_detect_corners_device", "sf_stereo_flow_defaults", "sf_stereo_correspondences_device", "sf_detector_defaults", "sf_get_features_and_descriptor", "sf_netvlad_load", "sf_netvlad_infer_device", "sf_netvlad_infer_batch_device", "sf_estimate_transform",
    "sf_estimate_transform_batch", "sf_verify_pairs", "sf_verify_pairs_device", "sf_verify_matches_device", "sf_find_matches_and_verify_device", "sf_compact_accepted_device",
    "sf_compact_accepted_device_async", "sf_step_issue", "sf_step_retire", "sf_memcpy_device_async", "sf_step_mirror", "sf_step_mirror_pair", "sf_step_mirror_streams", "sf_accept_stream_set", "sf_accept_stream_select", "sf_accept_stream_status", "sf_last_match_results", "sf_compact_accepted_indexed_device_async", "sf_compact_accepted_indexed_mirrored_device_async",
    "sf_debug_correspondences", "sf_debug_pass_state", "sf_debug_counters", "sf_debug_guided_points", "sf_debug_plan_workspace", "sf_pack_separators", "sf_comm_unique_id", "sf_comm_init", "sf_comm_destroy",
    "sf_allgather_separators", "sf_allgather_separators_device", "sf_allgather_bytes_device", "sf_nn_row_minima_device", "sf_nn_walk_device",
    "sf_get_features_and_descriptor_batch_device", "sf_prof_enable", "sf_prof_select", "sf_prof_reset", "sf_prof_get",
    "sf_kernel_name", "sf_stream_placement", "sf_streams_prepare",
]


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class SeparatorFinder:
    """One handle = one robot's separator-finder state on one GPU (single caller, like the
    reference's single-threaded geometry node)."""

    def __init__(self, params=None, device=0):
        self._L = load()
        self.params = _abi.copy_params(params) if params is not None else _abi.default_params()
        h = C.c_void_p()
        rc = self._L.sf_create(C.byref(self.params), device, C.byref(h))
        if rc != 0:
            raise SepfinderError(rc, (self._L.sf_last_error(None) or b"").decode())
        self._h = h
        self.device = device

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise SepfinderError(rc, (self._L.sf_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.sf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.sf_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        self._check(self._L.sf_synchronize(self._h))

    # -- NN stage (data_handler.py:166-209) ---------------------------------------------------
    def nn_append_local(self, desc):
        d = np.ascontiguousarray(desc, dtype=np.float64)
        d = d.reshape(-1, d.shape[-1]) if d.ndim > 1 else d.reshape(-1, self.params.netvlad_dimensions)
        self._check(self._L.sf_nn_append_local(self._h, _ptr(d), d.shape[0], d.shape[1]))

    def nn_append_received(self, desc):
        d = np.ascontiguousarray(desc, dtype=np.float64)
        d = d.reshape(-1, d.shape[-1]) if d.ndim > 1 else d.reshape(-1, self.params.netvlad_dimensions)
        self._check(self._L.sf_nn_append_received(self._h, _ptr(d), d.shape[0], d.shape[1]))

    def nn_append_local_device(self, dptr, n, dim):
        self._check(self._L.sf_nn_append_local_f32_device(self._h, C.c_void_p(dptr), n, dim))

    def nn_append_received_device(self, dptr, n, dim):
        self._check(self._L.sf_nn_append_received_f32_device(self._h, C.c_void_p(dptr), n, dim))

    def nn_append_local_f16_device(self, dptr, n, dim):
        """n x dim IEEE binary16 descriptors in device memory (converted exactly to the fp32 database rows)."""
        self._check(self._L.sf_nn_append_local_f16_device(self._h, C.c_void_p(dptr), n, dim))

    def nn_append_received_f16_device(self, dptr, n, dim):
        self._check(self._L.sf_nn_append_received_f16_device(self._h, C.c_void_p(dptr), n, dim))

    def nn_sizes(self):
        a, b = C.c_int32(), C.c_int32()
        self._check(self._L.sf_nn_sizes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def nn_mark_local_used(self, idx):
        self._check(self._L.sf_nn_mark_local_used(self._h, int(idx)))

    def nn_mark_other_used(self, idx):
        self._check(self._L.sf_nn_mark_other_used(self._h, int(idx)))

    def nn_ignore_pair(self, idx_local, idx_other):
        self._check(self._L.sf_nn_ignore_pair(self._h, int(idx_local), int(idx_other)))

    def nn_reset(self):
        self._check(self._L.sf_nn_reset(self._h))

    def nn_set_precision(self, nn_precision):
        self._check(self._L.sf_nn_set_precision(self._h, int(nn_precision)))
        self.params.nn_precision = int(nn_precision)

    def set_option(self, option, value):
        """Execution options of a live handle (_abi.SF_OPT_*); none changes any output byte."""
        self._check(self._L.sf_set_option(self._h, int(option), int(value)))

    def nn_find_matches(self, cap=None):
        n_l, _ = self.nn_sizes()
        if cap is None:
            cap = max(1, min(max(n_l, 1), self.params.netvlad_max_matches_nb))
        out = np.zeros(max(cap, 1), dtype=_abi.MATCH_DTYPE)
        n = C.c_int32()
        self._check(self._L.sf_nn_find_matches(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value]

    def nn_walk(self, row_min, row_arg, n_received, cap=None):
        """data_handler.py:191-205 on caller-provided per-row minima (host work; row-sharded NN of a multi-GPU node)."""
        d = np.ascontiguousarray(row_min, dtype=np.float64)
        a = np.ascontiguousarray(row_arg, dtype=np.int32)
        if cap is None:
            cap = max(1, min(max(d.size, 1), self.params.netvlad_max_matches_nb))
        out = np.zeros(max(cap, 1), dtype=_abi.MATCH_DTYPE)
        n = C.c_int32()
        self._check(self._L.sf_nn_walk(self._h, _ptr(d), _ptr(a), d.size, int(n_received), out.ctypes.data, cap,
                                       C.byref(n)))
        return out[: n.value]

    def nn_walk_device(self, d_row_min, d_row_arg, d_status, n_local, n_received, d_matches, cap, d_n_matches):
        """data_handler.py:191-205 on the device, on device-resident minima (raw pointers); asynchronous on the
        handle's stream.  d_matches: MATCH_DTYPE[cap], d_n_matches: int32[1] (device or pinned host memory)."""
        self._check(self._L.sf_nn_walk_device(self._h, C.c_void_p(d_row_min), C.c_void_p(d_row_arg),
                                              C.c_void_p(d_status) if d_status else None, int(n_local), int(n_received),
                                              C.c_void_p(d_matches), int(cap), C.c_void_p(d_n_matches)))

    def nn_row_minima_device(self, d_row_min, d_row_arg, d_status):
        """The NN kernels of this handle's local rows without the walk; results stay in device memory (raw pointers:
        float64[n_local], int32[n_local], int32[1]); asynchronous on the handle's stream."""
        self._check(self._L.sf_nn_row_minima_device(self._h, C.c_void_p(d_row_min), C.c_void_p(d_row_arg),
                                                    C.c_void_p(d_status)))

    def allgather_bytes_device(self, d_send, d_all, bytes_per_rank):
        self._check(self._L.sf_allgather_bytes_device(self._h, C.c_void_p(d_send), C.c_void_p(d_all), int(bytes_per_rank)))

    def nn_last_filter_dims(self):
        d = C.c_int32()
        self._check(self._L.sf_nn_last_filter_dims(self._h, C.byref(d)))
        return d.value

    def nn_last_row_minima(self):
        n_l, _ = self.nn_sizes()
        d = np.zeros(n_l, dtype=np.float64)
        i = np.zeros(n_l, dtype=np.int32)
        self._check(self._L.sf_nn_last_row_minima(self._h, _ptr(d), _ptr(i), n_l))
        return d, i

    # -- keyframe store ---------------------------------------------------------------------------
    def store_add_keyframe(self, feats):
        f = feats.c_struct()
        slot = C.c_int32()
        self._check(self._L.sf_store_add_keyframe(self._h, C.byref(f), C.byref(slot)))
        return slot.value

    def store_add_keyframes_device(self, n, rows, cols, d_desc, d_xyz, d_kp):
        first = C.c_int32()
        self._check(self._L.sf_store_add_keyframes_device(self._h, n, rows, cols, C.c_void_p(d_desc),
                                                          C.c_void_p(d_xyz), C.c_void_p(d_kp),
                                                          C.byref(first)))
        return first.value

    # -- feature extraction ------------------------------------------------
    def brief_set_pattern(self, tests):
        """tests: int8 [8 * bytes, 4] = (x1, y1, x2, y2) per descriptor bit."""
        t = np.ascontiguousarray(tests, dtype=np.int8).reshape(-1, 4)
        self._check(self._L.sf_brief_set_pattern(self._h, C.c_void_p(t.ctypes.data), t.shape[0] // 8))

    def brief_get_pattern(self):
        n = C.c_int32()
        buf = np.zeros((64 * 8, 4), np.int8)
        self._check(self._L.sf_brief_get_pattern(self._h, C.c_void_p(buf.ctypes.data), 64, C.byref(n)))
        return buf[:8 * n.value].copy()

    # -- NetVLAD inference -------------------------------------------------
    def netvlad_load(self, weights):
        """weights: dict of float32 numpy arrays in TensorFlow layouts -- conv_kernel[13] ([3, 3, Cin, Cout]),
        conv_bias[13], average_rgb [3], assignment [512, K], cluster_centers [512, K], wpca_kernel [512 K, pca_dim],
        wpca_bias [pca_dim]."""
        w = _abi.NetvladWeights()
        keep = []

        def ptr(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data
        for i in range(13):
            w.conv_kernel[i] = ptr(weights["conv_kernel"][i])
            w.conv_bias[i] = ptr(weights["conv_bias"][i])
        w.average_rgb = ptr(weights["average_rgb"])
        w.assignment = ptr(weights["assignment"])
        w.cluster_centers = ptr(weights["cluster_centers"])
        w.wpca_kernel = ptr(weights["wpca_kernel"])
        w.wpca_bias = ptr(weights["wpca_bias"])
        w.clusters = int(np.asarray(weights["assignment"]).shape[1])
        w.pca_dim = int(np.asarray(weights["wpca_bias"]).shape[0])
        self._check(self._L.sf_netvlad_load(self._h, C.byref(w)))

    def netvlad_infer_device(self, d_image_rgb, width, height, d_out, n_out):
        self._check(self._L.sf_netvlad_infer_device(self._h, C.c_void_p(d_image_rgb), width, height, C.c_void_p(d_out), n_out))

    def netvlad_infer_batch_device(self, d_images_rgb, n_images, width, height, d_out, n_out):
        """n_images images [H][W][3] float32 back to back on the device -> d_out [n_images][n_out] (data_handler.py:149-156)."""
        self._check(self._L.sf_netvlad_infer_batch_device(self._h, C.c_void_p(d_images_rgb), n_images, width, height,
                                                          C.c_void_p(d_out), n_out))

    def detect_corners_device(self, d_image, width, height, pitch, max_corners, quality_level, min_distance,
                              d_kpts_out, cap):
        """cv::goodFeaturesToTrack on the device; returns the number of corners found (<= cap are written)."""
        n = C.c_int32()
        self._check(self._L.sf_detect_corners_device(self._h, C.c_void_p(d_image), width, height, pitch, max_corners,
                                                     float(quality_level), float(min_distance), C.c_void_p(d_kpts_out),
                                                     cap, C.byref(n)))
        return n.value

    def stereo_correspondences_device(self, d_left, d_right, width, height, pitch, d_kpts, n, d_right_xy, d_status,
                                      d_right_x=None, d_err=None, params=None):
        """cv::calcOpticalFlowPyrLK + rtabmap's disparity gate on the device (asynchronous); device pointers (ints)."""
        self._check(self._L.sf_stereo_correspondences_device(
            self._h, C.c_void_p(d_left), C.c_void_p(d_right), width, height, pitch, C.c_void_p(d_kpts), n,
            C.byref(params) if params is not None else None, C.c_void_p(d_right_xy), C.c_void_p(d_status),
            C.c_void_p(d_right_x), C.c_void_p(d_err)))

    def get_features_and_descriptor(self, left, right, cam, det=None, flow=None):
        """GetFeatsAndDesc on host images (uint8 [h, w], same row stride): returns (descriptors [rows, bytes] uint8,
        kpts3D [rows, 3] float32, kpts [rows] KEYPOINT_DTYPE, slot of the keyframe in the device-resident store)."""
        left, right = np.asarray(left, np.uint8), np.asarray(right, np.uint8)
        if left.ndim != 2 or left.shape != right.shape or left.strides != right.strides or left.strides[1] != 1:
            raise ValueError("left / right must be 2-D uint8 images of one shape and row stride")
        h, w = left.shape
        cap = (det.max_features if det is not None else 1000)
        nbytes = self.brief_get_pattern().shape[0] // 8
        desc = np.zeros((cap, nbytes), np.uint8)
        xyz = np.zeros((cap, 3), np.float32)
        kp = np.zeros(cap, _abi.KEYPOINT_DTYPE)
        rows, slot = C.c_int32(), C.c_int32()
        self._check(self._L.sf_get_features_and_descriptor(
            self._h, C.c_void_p(left.ctypes.data), C.c_void_p(right.ctypes.data), w, h, left.strides[0], C.byref(cam),
            C.byref(det) if det is not None else None, C.byref(flow) if flow is not None else None,
            C.c_void_p(desc.ctypes.data), C.c_void_p(xyz.ctypes.data), C.c_void_p(kp.ctypes.data), cap, C.byref(rows),
            C.byref(slot)))
        n = min(rows.value, cap)
        return desc[:n].copy(), xyz[:n].copy(), kp[:n].copy(), slot.value

    def get_features_and_descriptor_batch_device(self, d_left, d_right, n_keyframes, width, height, pitch, image_stride,
                                                 cam, det=None, flow=None, d_rows_out=None, d_desc_out=None,
                                                 d_xyz_out=None, d_kpts_out=None):
        """n keyframes from device images (raw pointers) to n store slots in one launch sequence, no host wait; optional
        device outputs sized for n_keyframes x max_features rows.  Returns the first slot."""
        first = C.c_int32()
        self._check(self._L.sf_get_features_and_descriptor_batch_device(
            self._h, C.c_void_p(d_left), C.c_void_p(d_right), n_keyframes, width, height, pitch, int(image_stride),
            C.byref(cam), C.byref(det) if det is not None else None, C.byref(flow) if flow is not None else None,
            C.byref(first), C.c_void_p(d_rows_out), C.c_void_p(d_desc_out), C.c_void_p(d_xyz_out), C.c_void_p(d_kpts_out)))
        return first.value

    def extract_keyframe_device(self, d_left, width, height, pitch, d_kpts, d_right_x, d_status, n, cam,
                                d_desc_out=None, d_xyz_out=None, d_kpts_out=None, want_rows=True):
        """Device pointers (ints).  Returns (slot, rows kept or None)."""
        slot, rows = C.c_int32(), C.c_int32()
        self._check(self._L.sf_extract_keyframe_device(
            self._h, C.c_void_p(d_left), width, height, pitch, C.c_void_p(d_kpts), C.c_void_p(d_right_x),
            C.c_void_p(d_status), n, C.byref(cam), C.byref(slot), C.byref(rows) if want_rows else None,
            C.c_void_p(d_desc_out), C.c_void_p(d_xyz_out), C.c_void_p(d_kpts_out)))
        return slot.value, (rows.value if want_rows else None)

    def store_size(self):
        n = C.c_int32()
        self._check(self._L.sf_store_size(self._h, C.byref(n)))
        return n.value

    def store_clear(self):
        self._check(self._L.sf_store_clear(self._h))

    # -- verification -----------------------------------------------------------------------------
    def estimate_transform(self, f_from, f_to):
        res = np.zeros(1, dtype=_abi.RESULT_DTYPE)
        a, b = f_from.c_struct(), f_to.c_struct()
        self._check(self._L.sf_estimate_transform(self._h, C.byref(a), C.byref(b), res.ctypes.data))
        return res[0]

    def estimate_transform_batch(self, feats_from, feats_to):
        n = len(feats_from)
        res = np.zeros(n, dtype=_abi.RESULT_DTYPE)
        if n == 0:
            return res
        fa, ta = _abi.features_array(feats_from), _abi.features_array(feats_to)
        self._check(self._L.sf_estimate_transform_batch(self._h, fa, ta, n, res.ctypes.data))
        return res

    def verify_pairs(self, from_slots, to_slots):
        f = np.ascontiguousarray(from_slots, dtype=np.int32)
        t = np.ascontiguousarray(to_slots, dtype=np.int32)
        res = np.zeros(f.size, dtype=_abi.RESULT_DTYPE)
        self._check(self._L.sf_verify_pairs(self._h, _ptr(f), _ptr(t), f.size, _ptr(res)))
        return res

    def verify_pairs_device(self, d_from, d_to, n, d_out):
        self._check(self._L.sf_verify_pairs_device(self._h, C.c_void_p(d_from), C.c_void_p(d_to), n,
                                                   C.c_void_p(d_out)))

    def verify_matches_device(self, matches, slot_base_other, slot_base_local, d_out):
        """Verify the candidates of an NN query (structured array of MATCH_DTYPE, host) -> d_out (device)."""
        m = np.ascontiguousarray(matches, dtype=_abi.MATCH_DTYPE)
        self._check(self._L.sf_verify_matches_device(self._h, _ptr(m), m.size, slot_base_other, slot_base_local,
                                                     C.c_void_p(d_out)))
        return m.size

    def find_matches_and_verify_device(self, slot_base_other, slot_base_local, d_out, cap=None):
        """nn_find_matches + verify_matches_device as one call (speculative verification of the NN candidates
        while the host walks them, when the walk may return every local row); returns the matches, their
        results are written to d_out (device) asynchronously."""
        n_l, _ = self.nn_sizes()
        if cap is None:
            cap = max(1, min(max(n_l, 1), self.params.netvlad_max_matches_nb))
        out = np.zeros(max(cap, 1), dtype=_abi.MATCH_DTYPE)
        n = C.c_int32()
        self._check(self._L.sf_find_matches_and_verify_device(self._h, slot_base_other, slot_base_local, _ptr(out), cap,
                                                              C.byref(n), C.c_void_p(d_out) if d_out else None))
        return out[: n.value]

    # -- the caller's loop body as a begin / retire pair (find_separators.py:59-133) ----------------------------------
    def step_issue(self, slot_base_other, slot_base_local):
        """Queue one find-and-verify step (NN search, walk, verification of every returned candidate, all on the
        device); does not wait for any of it.  At most SF_OPT_STEP_DEPTH (default 6) steps in flight; with a mirror set
        (step_mirror / step_mirror_pair) as many as the mirror has buffers."""
        self._check(self._L.sf_step_issue(self._h, int(slot_base_other), int(slot_base_local)))

    def step_retire(self, copy=False):
        """The oldest step in flight: (matches, record_of_match, records, info).  The arrays are VIEWS of memory the
        handle owns (valid until the next step_retire) unless copy=True; records[record_of_match[i]] is the
        accepted result of match i, record_of_match[i] = -1 means its estimation failed."""
        r = _abi.StepResult()
        self._check(self._L.sf_step_retire(self._h, C.byref(r)))
        n, nr = r.n_matches, r.n_records

        def view(ptr, count, dtype):
            if not ptr or count <= 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dtype=dtype, count=count)
            return a.copy() if copy else a
        return (view(r.matches, n, _abi.MATCH_DTYPE), view(r.record_of_match, n, np.int32),
                view(r.records, nr, _abi.RESULT_DTYPE),
                {"n_matches": n, "n_records": nr, "n_accepted": r.n_accepted, "streamed": bool(r.streamed),
                 "d_records": r.d_records or 0})

    def memcpy_device_async(self, d_dst, d_src, nbytes, stream=None):
        """Device-to-device copy (raw pointers) on `stream` (a hipStream_t as an integer; None: the handle's stream), e.g. a
        retired step's d_records -> a send buffer."""
        self._check(self._L.sf_memcpy_device_async(self._h, C.c_void_p(d_dst), C.c_void_p(d_src), int(nbytes),
                                                   C.c_void_p(stream) if stream else None))

    def step_mirror(self, d_records2, d_counter, cap):
        """Second (device) destination of every accepted record + the caller's slot counter; (None, None, 0) removes it."""
        self._check(self._L.sf_step_mirror(self._h, C.c_void_p(d_records2) if d_records2 else None,
                                           C.c_void_p(d_counter) if d_counter else None, int(cap)))

    def step_mirror_pair(self, even, odd, cap):
        """Two alternating mirrors, (d_records2, d_counter) each: the first step issued after this call writes `even`."""
        self._check(self._L.sf_step_mirror_pair(self._h, C.c_void_p(even[0]), C.c_void_p(even[1]),
                                                C.c_void_p(odd[0]), C.c_void_p(odd[1]), int(cap)))

    def step_mirror_streams(self):
        """(stream_even, stream_odd) as integers: after step_mirror_pair, odd steps move to the handle's second stream."""
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self._L.sf_step_mirror_streams(self._h, C.byref(a), C.byref(b)))
        return a.value or 0, b.value or 0

    def last_match_results(self):
        """(d_results pointer, index pointer or None, n) of the last find_matches_and_verify_device call."""
        r, ix, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        self._check(self._L.sf_last_match_results(self._h, C.byref(r), C.byref(ix), C.byref(n)))
        return r.value, ix.value, n.value

    def compact_accepted_indexed_device_async(self, d_results, index, n, d_accepted, d_flags, d_count):
        self._check(self._L.sf_compact_accepted_indexed_device_async(
            self._h, C.c_void_p(d_results), C.c_void_p(index), n, C.c_void_p(d_accepted),
            C.c_void_p(d_flags) if d_flags else None, C.c_void_p(d_count)))

    def accept_stream_set(self, which, records, index, flags, cap, d_records2=None, d_counter=None):
        """Register block `which` (0 / 1) of the accepted-result stream: pinned host pointers (ints); optionally a
        second (device) destination of every record and a caller-owned device counter."""
        self._check(self._L.sf_accept_stream_set(self._h, which, C.c_void_p(records), C.c_void_p(index),
                                                 C.c_void_p(flags) if flags else None, cap,
                                                 C.c_void_p(d_records2) if d_records2 else None,
                                                 C.c_void_p(d_counter) if d_counter else None))

    def accept_stream_select(self, which):
        self._check(self._L.sf_accept_stream_select(self._h, which))

    def accept_stream_status(self):
        """(streamed, pairs) of the last sf_find_matches_and_verify_device."""
        s, n = C.c_int32(), C.c_int32()
        self._check(self._L.sf_accept_stream_status(self._h, C.byref(s), C.byref(n)))
        return bool(s.value), n.value

    def compact_accepted_indexed_mirrored_device_async(self, d_results, index, n, d_accepted, d_flags, d_count, d_accepted2,
                                                       d_flags2, d_count2):
        self._check(self._L.sf_compact_accepted_indexed_mirrored_device_async(
            self._h, C.c_void_p(d_results), C.c_void_p(index), n, C.c_void_p(d_accepted),
            C.c_void_p(d_flags) if d_flags else None, C.c_void_p(d_count), C.c_void_p(d_accepted2),
            C.c_void_p(d_flags2) if d_flags2 else None, C.c_void_p(d_count2)))

    def compact_accepted_device(self, d_results, n, d_accepted, d_flags=None):
        """Ordered device-side compaction of the accepted results; returns their number."""
        k = C.c_int32()
        self._check(self._L.sf_compact_accepted_device(self._h, C.c_void_p(d_results), n, C.c_void_p(d_accepted),
                                                       C.c_void_p(d_flags) if d_flags else None, C.byref(k)))
        return k.value

    def compact_accepted_device_async(self, d_results, n, d_accepted, d_flags, d_count):
        """The same without the synchronisation: the count (int32) is left on the device at d_count."""
        self._check(self._L.sf_compact_accepted_device_async(self._h, C.c_void_p(d_results), n, C.c_void_p(d_accepted),
                                                             C.c_void_p(d_flags) if d_flags else None, C.c_void_p(d_count)))

    def debug_pass_state(self, pair, which_pass):
        """(T [3][4] float32, is_null, inliers, matches) of one pass of pair `pair` of the last verification."""
        T = np.zeros(12, dtype=np.float32)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.sf_debug_pass_state(self._h, pair, which_pass, T.ctypes.data, C.byref(a), C.byref(b), C.byref(c)))
        return T.reshape(3, 4), a.value, b.value, c.value

    def debug_guided_points(self, pair, kcap=4096):
        a = np.zeros(kcap, dtype=np.uint64); b = np.zeros(kcap, dtype=np.uint64)
        k = C.c_int32()
        self._check(self._L.sf_debug_guided_points(self._h, pair, a.ctypes.data, b.ctypes.data, C.byref(k)))
        return a[: k.value].copy(), b[: k.value].copy()

    def debug_counters(self, n=8):
        out = np.zeros(n, dtype=np.uint64)
        self._check(self._L.sf_debug_counters(self._h, out.ctypes.data, n))
        return out

    def debug_correspondences(self, pair, which_pass, cap=4096):
        cf = np.zeros(cap, dtype=np.uint16)
        ct = np.zeros(cap, dtype=np.uint16)
        n = C.c_int32()
        self._check(self._L.sf_debug_correspondences(self._h, pair, which_pass, cf.ctypes.data,
                                                     ct.ctypes.data, cap, C.byref(n)))
        return cf[: n.value].copy(), ct[: n.value].copy()

    # -- multi-GPU exchange (RCCL through the C-ABI) --------------------------------------------------
    def comm_init(self, unique_id, rank, world):
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        self._check(self._L.sf_comm_init(self._h, uid.ctypes.data, rank, world))

    def comm_destroy(self):
        self._check(self._L.sf_comm_destroy(self._h))

    def allgather_separators(self, d_local, n_local, d_all, cap_per_rank, world):
        counts = np.zeros(world, dtype=np.int32)
        self._check(self._L.sf_allgather_separators(self._h, C.c_void_p(d_local), n_local, C.c_void_p(d_all),
                                                    cap_per_rank, counts.ctypes.data))
        return counts

    # -- measurement ------------------------------------------------------------------------------
    def allgather_separators_device(self, d_send, d_all, cap_per_rank):
        """One-collective exchange on device buffers with a device-stamped count (see include/sepfinder.h)."""
        self._check(self._L.sf_allgather_separators_device(self._h, C.c_void_p(d_send), C.c_void_p(d_all), cap_per_rank))

    def prof_enable(self, on=True):
        self._check(self._L.sf_prof_enable(self._h, int(on)))

    def prof_select(self, names=None):
        """Bracket only the named kernels (sf_kernel_name strings) while profiling is on; None = all."""
        mask = 0xFFFFFFFF
        if names is not None:
            mask = 0
            for k in range(_abi.SF_K_COUNT):
                if self._L.sf_kernel_name(k).decode() in names:
                    mask |= 1 << k
        self._check(self._L.sf_prof_select(self._h, mask))

    def stream_placement(self):
        """One line on where the step pipeline's streams sit on the dispatch pipes (include/sf_experimental.h)."""
        buf = C.create_string_buffer(512)
        self._check(self._L.sf_stream_placement(self._h, buf, 512))
        return buf.value.decode()

    def streams_prepare(self):
        """Run the stream placement measurement now (60-100 ms; otherwise inside the first overlapped step)."""
        self._check(self._L.sf_streams_prepare(self._h))

    def prof_reset(self):
        self._check(self._L.sf_prof_reset(self._h))

    def prof_get(self):
        """{kernel name: (launches, total_ms)} measured with hipEvents on the handle's stream."""
        out = {}
        for k in range(_abi.SF_K_COUNT):
            n, ms = C.c_int64(), C.c_double()
            self._check(self._L.sf_prof_get(self._h, k, C.byref(n), C.byref(ms)))
            out[self._L.sf_kernel_name(k).decode()] = (n.value, ms.value)
        return out


def comm_unique_id():
    """ncclUniqueId bytes (created by one rank, distributed by the host side)."""
    L = load()
    out = np.zeros(128, dtype=np.uint8)
    rc = L.sf_comm_unique_id(out.ctypes.data, 128)
    if rc != 0:
        raise SepfinderError(rc, "sf_comm_unique_id (is librccl.so loadable?)")
    return out.tobytes()


def pack_separators(results, robot_from, robot_to, kf_from, kf_to, frame_from, frame_to):
    """ReceiveSeparators.srv rows (one per result) as a structured array of SEPARATOR_DTYPE."""
    L = load()
    res = np.ascontiguousarray(results, dtype=_abi.RESULT_DTYPE)
    n = res.size
    arrs = [np.ascontiguousarray(a, dtype=np.int16) for a in (kf_from, kf_to, frame_from, frame_to)]
    for a in arrs:
        if a.size != n:
            raise ValueError("id arrays must have one entry per result")
    for v in (robot_from, robot_to):
        if not -128 <= int(v) <= 127:
            raise ValueError("robot ids are int8 on the wire (ReceiveSeparators.srv:1-2)")
    out = np.zeros(n, dtype=_abi.SEPARATOR_DTYPE)
    rc = L.sf_pack_separators(_ptr(res), n, int(robot_from), int(robot_to), _ptr(arrs[0]), _ptr(arrs[1]),
                              _ptr(arrs[2]), _ptr(arrs[3]), _ptr(out))
    if rc != 0:
        raise SepfinderError(rc, "sf_pack_separators")
    return out
