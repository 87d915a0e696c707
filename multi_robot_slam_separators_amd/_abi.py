"""ctypes mirror of include/sepfinder.h (POD structs + constants of the C-ABI boundary).

Layouts follow the reference's wire types (PKG = ros_ws/src/multi_robot_separators):
PKG/msg/Descriptors.msg:1-3, KeyPoint3DVec.msg:1-2, KeyPointVec.msg:1-2,
PKG/srv/EstTransform.srv:1-9, PKG/srv/ReceiveSeparators.srv:1-10.
"""
import ctypes as C

import numpy as np

SF_OK, SF_EINVAL, SF_EHIP, SF_ENOMEM, SF_ERANGE, SF_ENODEV, SF_ERCCL = range(7)
SF_MAX_FEATURES = 32767
SF_MAX_DESC_BYTES = 64
SF_MAX_DESC_BYTES_F32 = 512

(SF_K_MATCH, SF_K_RANSAC1, SF_K_GUIDED, SF_K_RANSAC2, SF_K_NN, SF_K_NN_SELECT, SF_K_NN_FILTER,
 SF_K_NN_REFINE, SF_K_FUSED, SF_K_NN_WALK, SF_K_BA, SF_K_COUNT) = range(12)
(SF_OPT_MATCH_MFMA, SF_OPT_FUSED, SF_OPT_OVERLAP, SF_OPT_CHAIN_WAVES, SF_OPT_DEBUG_CORR, SF_OPT_NN_FULL_FILTER,
 SF_OPT_STEP_OVERLAP, SF_OPT_STEP_SPLIT, SF_OPT_STEP_DEPTH, SF_OPT_STEP_LANES, SF_OPT_STEP_DEVICE_WALK,
 SF_OPT_STEP_SPECULATE) = range(12)      # sf_set_option

STATUS_NAMES = {0: "SF_OK", 1: "SF_EINVAL", 2: "SF_EHIP", 3: "SF_ENOMEM", 4: "SF_ERANGE", 5: "SF_ENODEV", 6: "SF_ERCCL"}


class Params(C.Structure):
    _fields_ = [
        ("netvlad_distance", C.c_double),
        ("netvlad_dimensions", C.c_int32),
        ("netvlad_max_matches_nb", C.c_int32),
        ("nn_precision", C.c_int32),
        ("min_inliers", C.c_int32),
        ("inlier_distance", C.c_float),
        ("iterations", C.c_int32),
        ("refine_iterations", C.c_int32),
        ("refine_sigma", C.c_double),
        ("estimation_type", C.c_int32),
        ("nndr", C.c_float),
        ("guess_win_size", C.c_int32),
        ("ransac_adaptive_stop", C.c_int32),
        ("max_sample_checks", C.c_int32),
        ("seed", C.c_uint64),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("image_width", C.c_int32),
        ("image_height", C.c_int32),
        ("local_transform", C.c_float * 12),
        ("store_capacity", C.c_int32),
        ("max_features", C.c_int32),
        ("desc_bytes", C.c_int32),
        ("pnp_reproj_error", C.c_float),
        ("pnp_flags", C.c_int32),
        ("pnp_refine_iterations", C.c_int32),
        ("bundle_adjustment", C.c_int32),
        ("ba_iterations", C.c_int32),
        ("ba_robust_kernel_delta", C.c_float),
        ("ba_pixel_variance", C.c_float),
        ("stereo_baseline", C.c_float),
        ("force_3dof", C.c_int32),
        ("forward_est_only", C.c_int32),
        ("desc_type", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class Keypoint(C.Structure):
    _fields_ = [
        ("x", C.c_float),
        ("y", C.c_float),
        ("size", C.c_float),
        ("angle", C.c_float),
        ("response", C.c_float),
        ("octave", C.c_int32),
        ("class_id", C.c_int32),
    ]


KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
assert KEYPOINT_DTYPE.itemsize == C.sizeof(Keypoint) == 28


class StereoCamera(C.Structure):
    """sf_stereo_camera (include/sepfinder.h): the stereo model of sf_extract_keyframe_device."""
    _fields_ = [
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("cx_right", C.c_float), ("baseline", C.c_float),
        ("local_transform", C.c_float * 12),
        ("min_depth", C.c_float), ("max_depth", C.c_float),
    ]


def stereo_camera(fx, fy, cx, cy, baseline, cx_right=0.0, local_transform=None, min_depth=0.0, max_depth=0.0):
    cam = StereoCamera()
    cam.fx, cam.fy, cam.cx, cam.cy, cam.cx_right, cam.baseline = fx, fy, cx, cy, cx_right, baseline
    lt = np.eye(4, dtype=np.float32)[:3] if local_transform is None else np.asarray(local_transform, np.float32).reshape(3, 4)
    for i, v in enumerate(lt.ravel()):
        cam.local_transform[i] = float(v)
    cam.min_depth, cam.max_depth = min_depth, max_depth
    return cam


class StereoFlowParams(C.Structure):
    """sf_stereo_flow_params (include/sepfinder.h): cv::calcOpticalFlowPyrLK + disparity gate of the stereo correspondence."""
    _fields_ = [
        ("win_width", C.c_int32), ("win_height", C.c_int32), ("max_level", C.c_int32), ("iterations", C.c_int32),
        ("epsilon", C.c_double), ("min_disparity", C.c_float), ("max_disparity", C.c_float),
        ("min_eig_threshold", C.c_float),
    ]


def stereo_flow_params(win_width=15, win_height=3, max_level=5, iterations=30, epsilon=0.01, min_disparity=0.5,
                       max_disparity=128.0, min_eig_threshold=1e-4):
    """rtabmap's Stereo/* defaults (what sf_stereo_flow_defaults fills)."""
    p = StereoFlowParams()
    p.win_width, p.win_height, p.max_level, p.iterations = win_width, win_height, max_level, iterations
    p.epsilon, p.min_disparity, p.max_disparity, p.min_eig_threshold = epsilon, min_disparity, max_disparity, min_eig_threshold
    return p


class DetectorParams(C.Structure):
    """sf_detector_params (include/sepfinder.h): cv::goodFeaturesToTrack's arguments in sf_get_features_and_descriptor."""
    _fields_ = [("max_features", C.c_int32), ("quality_level", C.c_double), ("min_distance", C.c_double)]


def detector_params(max_features=1000, quality_level=0.001, min_distance=3.0):
    """rtabmap's Vis/MaxFeatures, GFTT/QualityLevel, GFTT/MinDistance defaults (what sf_detector_defaults fills)."""
    p = DetectorParams()
    p.max_features, p.quality_level, p.min_distance = max_features, quality_level, min_distance
    return p


class NetvladWeights(C.Structure):
    """sf_netvlad_weights (include/sepfinder.h): host pointers to the NetVLAD network's weights, TensorFlow layouts."""
    _fields_ = [
        ("conv_kernel", C.c_void_p * 13), ("conv_bias", C.c_void_p * 13),
        ("average_rgb", C.c_void_p), ("assignment", C.c_void_p), ("cluster_centers", C.c_void_p),
        ("wpca_kernel", C.c_void_p), ("wpca_bias", C.c_void_p),
        ("clusters", C.c_int32), ("pca_dim", C.c_int32),
    ]


VGG16_CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512),
               (512, 512), (512, 512), (512, 512), (512, 512)]          # (Cin, Cout) of conv1_1 ... conv5_3


class Features(C.Structure):
    _fields_ = [
        ("desc", C.c_void_p),
        ("rows", C.c_uint16),
        ("cols", C.c_uint16),
        ("xyz", C.c_void_p),
        ("n3d", C.c_int32),
        ("kpts", C.c_void_p),
        ("nkp", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("position", C.c_double * 3),
        ("orientation", C.c_double * 4),
        ("covariance", C.c_double * 36),
        ("inliers", C.c_int32),
        ("matches", C.c_int32),
        ("inliers_pass1", C.c_int32),
        ("matches_pass1", C.c_int32),
        ("success", C.c_uint8),
        ("pass1_success", C.c_uint8),
        ("pass2_guided", C.c_uint8),
        ("pad", C.c_uint8 * 5),
    ]


RESULT_DTYPE = np.dtype(
    [("position", "<f8", (3,)), ("orientation", "<f8", (4,)), ("covariance", "<f8", (36,)),
     ("inliers", "<i4"), ("matches", "<i4"), ("inliers_pass1", "<i4"), ("matches_pass1", "<i4"),
     ("success", "u1"), ("pass1_success", "u1"), ("pass2_guided", "u1"), ("pad", "u1", (5,))]
)
assert RESULT_DTYPE.itemsize == C.sizeof(Result) == 368


class Separator(C.Structure):
    _fields_ = [
        ("robot_from_id", C.c_int8),
        ("robot_to_id", C.c_int8),
        ("kf_id_from", C.c_int16),
        ("kf_id_to", C.c_int16),
        ("frame_id_from", C.c_int16),
        ("frame_id_to", C.c_int16),
        ("transform_est_success", C.c_uint8),
        ("pad", C.c_uint8 * 5),
        ("position", C.c_double * 3),
        ("orientation", C.c_double * 4),
        ("covariance", C.c_double * 36),
    ]


SEPARATOR_DTYPE = np.dtype(
    [("robot_from_id", "i1"), ("robot_to_id", "i1"), ("kf_id_from", "<i2"), ("kf_id_to", "<i2"),
     ("frame_id_from", "<i2"), ("frame_id_to", "<i2"), ("transform_est_success", "u1"),
     ("pad", "u1", (5,)), ("position", "<f8", (3,)), ("orientation", "<f8", (4,)),
     ("covariance", "<f8", (36,))]
)
assert SEPARATOR_DTYPE.itemsize == C.sizeof(Separator) == 360


class Match(C.Structure):
    _fields_ = [("idx_local", C.c_int32), ("idx_other", C.c_int32), ("distance", C.c_double)]


MATCH_DTYPE = np.dtype([("idx_local", "<i4"), ("idx_other", "<i4"), ("distance", "<f8")])
assert MATCH_DTYPE.itemsize == C.sizeof(Match) == 16


class StepResult(C.Structure):
    """sf_step_result (include/sepfinder.h): what sf_step_retire hands back; the pointers belong to the handle."""
    _fields_ = [("matches", C.c_void_p), ("record_of_match", C.c_void_p), ("records", C.c_void_p),
                ("n_matches", C.c_int32), ("n_records", C.c_int32), ("n_accepted", C.c_int32), ("streamed", C.c_int32),
                ("d_records", C.c_void_p)]


SF_ABI_VERSION = 6      # include/sepfinder.h


def default_params() -> Params:
    """Defaults: multi_robot_separators.launch:19-23 for the reference's own knobs; rtabmap
    compiled-in defaults [upstream, SURVEY.md section 9] for the Vis/* values; estimation type
    0 (3D->3D) as BASELINE.json's north_star names.  Mirrors sf_default_params()."""
    p = Params()
    p.netvlad_distance = 0.13
    p.netvlad_dimensions = 128
    p.netvlad_max_matches_nb = 20
    p.nn_precision = 1
    p.min_inliers = 5
    p.inlier_distance = 0.1
    p.iterations = 300
    p.refine_iterations = 5
    p.refine_sigma = 3.0
    p.estimation_type = 0
    p.nndr = 0.6
    p.guess_win_size = 20
    p.ransac_adaptive_stop = 1
    p.max_sample_checks = 1000
    p.seed = 12345
    p.fx = p.fy = 0.0
    p.cx = p.cy = 0.0
    p.image_width = p.image_height = 0
    for i, v in enumerate([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]):
        p.local_transform[i] = float(v)
    p.store_capacity = 1024
    p.max_features = 512
    p.desc_bytes = 32
    p.pnp_reproj_error = 2.0
    p.pnp_flags = 0
    p.pnp_refine_iterations = 0
    p.bundle_adjustment = 0
    p.ba_iterations = 20
    p.ba_robust_kernel_delta = 8.0
    p.ba_pixel_variance = 1.0
    p.stereo_baseline = 0.0
    p.force_3dof = 0
    p.forward_est_only = 1
    p.desc_type = 0
    p.reserved0 = 0
    return p


def copy_params(p: Params) -> Params:
    q = Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(Params))
    return q


class FeatureArrays:
    """One keyframe's geometric features as numpy arrays, with a ctypes `Features` view that
    borrows them (zero copy, like descriptorsFromROS at MsgConversion.cpp:113-116)."""

    def __init__(self, desc, xyz, kpts):
        desc = np.asarray(desc)
        if desc.dtype == np.float32:       # float32 descriptor rows (sf_params.desc_type 1): handed over as their bytes
            desc = np.ascontiguousarray(desc).view(np.uint8).reshape(desc.shape[0], 4 * desc.shape[1])
        self.desc = np.ascontiguousarray(desc, dtype=np.uint8)
        if self.desc.ndim != 2:
            raise ValueError("desc must be rows x cols uint8 (or rows x dims float32)")
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        kpts = np.asarray(kpts)
        if kpts.dtype != KEYPOINT_DTYPE:
            raise ValueError("kpts must have KEYPOINT_DTYPE")
        self.kpts = np.ascontiguousarray(kpts)

    def c_struct(self) -> Features:
        f = Features()
        f.rows, f.cols = self.desc.shape
        f.desc = self.desc.ctypes.data if self.desc.size else None
        f.n3d = self.xyz.shape[0]
        f.xyz = self.xyz.ctypes.data if self.xyz.size else None
        f.nkp = self.kpts.shape[0]
        f.kpts = self.kpts.ctypes.data if self.kpts.size else None
        return f


def features_array(feats):
    """ctypes array of Features for a list of FeatureArrays (keeps borrowing the numpy data)."""
    arr = (Features * len(feats))()
    for i, f in enumerate(feats):
        arr[i] = f.c_struct()
    return arr
