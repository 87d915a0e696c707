/*
 * sepfinder.h -- C-ABI boundary of the MI355X-native inter-robot separator finder.
 *
 * This is the drop-in surface for ONE hot path of bramtoula/multi_robot_SLAM_separators
 * (SURVEY.md section 8): NetVLAD nearest-neighbour candidate search + binary local-descriptor
 * matching + RANSAC relative pose (3D-3D, or 3D-2D PnP with estimation_type = 1).  Everything is `extern "C"`, plain pointers and sizes,
 * no torch / ROS / OpenCV types.  All citations are relative to the reference tree, with
 * PKG = ros_ws/src/multi_robot_separators.
 *
 * Reference interface each group of entry points replaces:
 *   sf_nn_*            PKG/scripts/data_handler.py:166-209   DataHandler.find_matches
 *                      PKG/scripts/data_handler.py:297-301   find_matches_service (descriptor append)
 *                      PKG/scripts/data_handler.py:402-408,437-438  mask bookkeeping
 *                      wire: PKG/srv/FindMatches.srv:1 (float64[] new_netvlad_descriptors)
 *   sf_store_*         PKG/scripts/data_handler.py:268,421-422  geometric_feats cache of one keyframe's
 *                      (descriptors, kpts3D, kpts); layouts PKG/msg/Descriptors.msg:1-3,
 *                      KeyPoint3DVec.msg:1-2, KeyPointVec.msg:1-2, PKG/src/MsgConversion.cpp:8-42,113-116
 *   sf_estimate_*      PKG/src/stereoCamGeometricTools.cpp:122-178  estimateTransformation service
 *                      (PKG/srv/EstTransform.srv:1-9), which drives
 *                      PKG/src/myRegistration.cpp:225-303 and PKG/src/myRegistrationVis.cpp:441-1410
 *   sf_verify_*_device, sf_find_matches_and_verify_device, sf_compact_accepted_device
 *                      PKG/scripts/find_separators.py:59-133  the caller's loop body (find_matches, then one
 *                      estimate_transformation per returned candidate, then the accepted separators) for
 *                      keyframes that already live in the handle's device store
 *   sf_result          geometry_msgs/PoseWithCovariance + bool success (EstTransform.srv:8-9),
 *                      packed as PKG/src/MsgConversion.cpp:61-64,71-81 do
 *   sf_separator       one row of PKG/srv/ReceiveSeparators.srv:1-10 (what the back-end consumes,
 *                      PKG/src/factorGraph.cpp:98-118)
 *
 * Ownership: every input pointer is BORROWED for the duration of the call; every output is
 * written into CALLER-allocated memory; the handle owns all device memory.  No allocation
 * crosses the ABI.  A handle is single-caller (not thread-safe), like the reference's
 * single-threaded geometry node (stereoCamGeometricTools.cpp:212).
 *
 * Errors: every function returns an int status.  A failed pose estimation is NOT an error
 * (success=0, zero pose, covariance as the reference leaves it).  Size mismatches on which the
 * reference UASSERT-aborts (myRegistrationVis.cpp:482-483,859-860,878-880) return SF_EINVAL.
 */
#ifndef SEPFINDER_H
#define SEPFINDER_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: sf_params.reserved[5] became bundle_adjustment / ba_* / stereo_baseline, sf_find_matches_and_verify_device accepts
      d_out = NULL, sf_debug_correspondences needs SF_OPT_DEBUG_CORR, sf_step_* added, the accepted-result stream and the
      indexed compactions moved to sf_experimental.h.  A binding checks sf_abi_version() against the header it was
      written for (multi_robot_slam_separators_amd/lib.py does).                                                   */
/* 3: sf_params grew force_3dof / forward_est_only (appended), sf_nn_row_minima_device, sf_allgather_bytes_device,
      sf_netvlad_infer_batch_device, sf_get_features_and_descriptor_batch_device added.                                                                          */
/* 4: sf_step_mirror_pair, sf_step_mirror_streams, SF_OPT_STEP_SPLIT added (nothing existing changed).                */
/* 5: sf_step_issue no longer waits for the device (SF_OPT_STEP_DEVICE_WALK, _DEPTH, _LANES), sf_nn_walk_device added; 
      sf_step_result pointers stay valid until the next sf_step_retire; sf_params grew desc_type (appended).         */
/* 6: SF_K_BA added to the kernel ids of sf_prof_get (SF_K_COUNT 10 -> 11): the bundle adjustment is a launch of its
      own; sf_step_issue refuses more steps in flight than a mirror has buffers for.                                */
#define SF_ABI_VERSION 6

/* ---- status codes ---------------------------------------------------------------------- */
enum {
  SF_OK = 0,
  SF_EINVAL = 1,   /* bad argument / size mismatch (reference: UASSERT abort)            */
  SF_EHIP = 2,     /* HIP runtime error; text via sf_last_error                          */
  SF_ENOMEM = 3,   /* host or device allocation failed                                   */
  SF_ERANGE = 4,   /* index out of range / capacity exceeded / IDL limit exceeded         */
  SF_ENODEV = 5,   /* no GPU visible: the product path has NO CPU fallback                */
  SF_ERCCL = 6     /* RCCL error (or librccl.so not loadable); text via sf_last_error        */
};

/* ---- limits inherited from the reference IDL --------------------------------------------- */
#define SF_MAX_FEATURES   32767  /* KeyPoint3DVec.msg:1 / KeyPointVec.msg:1  `int16 size`   */
#define SF_MAX_DESC_BYTES 64     /* 512-bit binary descriptors (BASELINE.json configs[4])   */
#define SF_MAX_DESC_BYTES_F32 512 /* float32 descriptors (desc_type 1): 64 or 128 dimensions  */

/* ---- parameters -------------------------------------------------------------------------- */
typedef struct sf_params {
  /* NetVLAD NN stage: data_handler.py:96-100, defaults multi_robot_separators.launch:19-22 */
  double  netvlad_distance;        /* 0.13 ; accept iff dist <  this (data_handler.py:202)  */
  int32_t netvlad_dimensions;      /* 128                                                   */
  int32_t netvlad_max_matches_nb;  /* 20                                                    */
  int32_t nn_precision;            /* 0 = fp32 MFMA ranks every column, f64 re-evaluation of the
                                          row minima (all rows exact);
                                      1 = fp16 MFMA FILTER with a rigorous error band + f64
                                          evaluation of every surviving (row, column): identical
                                          matches; rows whose minimum is >= netvlad_distance
                                          report +inf in sf_nn_last_row_minima
                                          (BASELINE configs[4]: fp16 NetVLAD on MFMA)           */
  /* Registration: myRegistrationVis.cpp:52-71 reads rtabmap's compiled-in defaults [upstream];
     stereoCamGeometricTools.cpp:87 overrides only Vis/MinInliers                           */
  int32_t min_inliers;             /* 5    separators_min_inliers -> Vis/MinInliers          */
  float   inlier_distance;         /* 0.1  Vis/InlierDistance (m)                            */
  int32_t iterations;              /* 300  Vis/Iterations                                    */
  int32_t refine_iterations;       /* 5    Vis/RefineIterations                              */
  double  refine_sigma;            /* 3.0  refineModelSigma (rtabmap util3d_registration)    */
  int32_t estimation_type;         /* 0 = 3D->3D (north_star; myRegistrationVis.cpp:1113-1152),
                                      1 = 3D->2D PnP (:1055-1112; rtabmap's compiled-in default),
                                      2 = epipolar is NOT implemented: sf_create -> SF_EINVAL   */
  float   nndr;                    /* 0.6  Vis/CorNNDR                                       */
  int32_t guess_win_size;          /* 20   Vis/CorGuessWinSize (px); 0 disables guided pass  */
  int32_t ransac_adaptive_stop;    /* 1 = PCL RandomSampleConsensus adaptive k (p=0.99);
                                      0 = evaluate exactly iterations+1 hypotheses           */
  int32_t max_sample_checks;       /* 1000 PCL SampleConsensusModel::max_sample_checks_      */
  uint64_t seed;                   /* sampler key (PCL uses a fixed mt19937 seed 12345)      */
  /* Camera model `cam_` used for BOTH frames (stereoCamGeometricTools.cpp:76,142-143)       */
  double  fx, fy, cx, cy;          /* left().K()                                             */
  int32_t image_width, image_height; /* left().imageSize(); 0 => uncalibrated => no guided pass */
  float   local_transform[12];     /* left().localTransform(): base->optical, row-major 3x4  */
  /* Capacity hints (device memory is grown on demand; these avoid regrowth)                 */
  int32_t store_capacity;          /* keyframes                                              */
  int32_t max_features;            /* per keyframe (rounded up to a multiple of 64)          */
  int32_t desc_bytes;              /* descriptor bytes per feature, 1..64 (32 = ORB/BRIEF)   */
  /* PnP branch (estimation_type = 1): myRegistrationVis.cpp:59-61,1083-1085 [upstream defaults]  */
  float   pnp_reproj_error;        /* 2.0  Vis/PnPReprojError (px); inlier iff error <= this  */
  int32_t pnp_flags;               /* 0    Vis/PnPFlags: only 0 (cv::SOLVEPNP_ITERATIVE)      */
  int32_t pnp_refine_iterations;   /* 0    Vis/PnPRefineIterations (rtabmap util3d::solvePnPRansac
                                           re-solve / re-select rounds, 3-sigma threshold)        */
  /* Two-view bundle adjustment after each pass's motion estimate (myRegistrationVis.cpp:1192-1370; rtabmap
     Vis/BundleAdjustment, upstream default 1 when built with g2o): the inliers' 3D points (from-frame) and the pose of
     the "to" camera are refined against both frames' keypoints, the "from" pose fixed, Huber kernel; words whose
     reprojection stays beyond the kernel width are removed from the inliers (sbaOutliers, :1314-1330) and the
     min_inliers test is repeated (:1331-1336).                                                              */
  int32_t bundle_adjustment;       /* 0 (default here; north_star's path) = off, 1 = on (needs a calibrated camera) */
  int32_t ba_iterations;           /* 20   Optimizer/Iterations                                             */
  float   ba_robust_kernel_delta;  /* 8.0  g2o/RobustKernelDelta (pixels)                                   */
  float   ba_pixel_variance;       /* 1.0  g2o/PixelVariance                                                */
  float   stereo_baseline;         /* metres; > 0 adds the stereo (disparity) residual of points with depth  */
  /* Reg/Force3DoF (myRegistration.cpp:245-248,269-276; myRegistrationVis.cpp:1100-1102,1141-1143; rtabmap default
     false): guess and estimates are reduced to (x, y, yaw) -- Transform::to3DoF [upstream] = Transform(x, y, 0, 0, 0,
     yaw) with yaw = atan2(r21, r11), computed here as the rotation (r11, r21) / |(r11, r21)| about z.            */
  int32_t force_3dof;              /* 0 */
  /* Vis/ForwardEstOnly (myRegistrationVis.cpp:936-978,1155-1197,1369,1376-1394; rtabmap default true).  0: every pass
     also estimates to -> from (the frames' roles swapped, each direction behind its own gate); the pass's transform is
     interpolate(0.5) of the forward estimate and the inverse of the backward one, its covariance their mean, inliers /
     matches the union of both directions' ids.  With bundle_adjustment the forward transform is adjusted over the union
     of the inliers and the backward transform is dropped (:1369).  Both estimators; runs on the stage kernels.   */
  int32_t forward_est_only;        /* 1 */
  /* Descriptor type of every keyframe of this handle (rtabmap Vis/FeatureType is fixed per run).  0 = binary rows
     (CV_8U: BRIEF / ORB, Hamming distance -- the only kind the reference's own wire carries, MsgConversion.cpp:113-129);
     1 = float32 rows (SURF / SIFT: `desc` points at rows x dims float32, `cols` = 4 * dims with dims = 64 or 128):
     brute-force kNN-2 on the L2 distance with the same NNDR / uniqueness / window rules -- squared distances in the
     global matching (myRegistrationVis.cpp:839-854 -> VWDictionary::addNewWords [upstream]), cv::BFMatcher(NORM_L2)
     distances in the guided matching (:739-749).  north_star's "ORB/SURF ... Hamming/L2 matching".               */
  int32_t desc_type;               /* 0 */
  int32_t reserved0;               /* 0 (keeps the struct a multiple of 8 bytes)                               */
} sf_params;

/* ---- wire layouts ------------------------------------------------------------------------ */
/* rtabmap_ros/KeyPoint as carried by KeyPointVec.msg (fields per MsgConversion.cpp:50-56)   */
typedef struct sf_keypoint {
  float   x, y;        /* pt.x, pt.y (pixels) */
  float   size;
  float   angle;
  float   response;
  int32_t octave;
  int32_t class_id;
} sf_keypoint;         /* 28 bytes */

/* One keyframe's geometric features = (Descriptors, KeyPoint3DVec, KeyPointVec)            */
typedef struct sf_features {
  const uint8_t*     desc;      /* rows x cols, row-major (Descriptors.msg, zero-copy view
                                   exactly as MsgConversion.cpp:113-116)                     */
  uint16_t           rows;      /* Descriptors.rows */
  uint16_t           cols;      /* Descriptors.cols (bytes per descriptor) */
  const float*       xyz;       /* n3d x 3 float32, base frame (KeyPoint3DVec.kpts3DVec)     */
  int32_t            n3d;       /* KeyPoint3DVec.size ; must be 0 or rows                    */
  const sf_keypoint* kpts;      /* nkp entries (KeyPointVec.kptsVec)                         */
  int32_t            nkp;       /* KeyPointVec.size ; must equal rows (or 0 when rows==0)    */
} sf_features;

/* EstTransform.srv response + the RegistrationInfo fields the node keeps (info_)           */
typedef struct sf_result {
  double  position[3];     /* geometry_msgs/Pose.position                                   */
  double  orientation[4];  /* quaternion x,y,z,w (w >= 0, tf::poseEigenToMsg)               */
  double  covariance[36];  /* row-major 6x6, as memcpy'd by MsgConversion.cpp:61-64         */
  int32_t inliers;         /* RegistrationInfo.inliers of pass 2                            */
  int32_t matches;         /* RegistrationInfo.matches of pass 2                            */
  int32_t inliers_pass1;
  int32_t matches_pass1;
  uint8_t success;         /* !result2.isNull()  (stereoCamGeometricTools.cpp:168-175)      */
  uint8_t pass1_success;
  uint8_t pass2_guided;    /* 1 if pass 2 took the guess-guided branch                      */
  uint8_t pad[5];
} sf_result;               /* 368 bytes */

/* One row of ReceiveSeparators.srv, the record all-gathered across GPUs                    */
typedef struct sf_separator {
  int8_t  robot_from_id, robot_to_id;
  int16_t kf_id_from, kf_id_to;
  int16_t frame_id_from, frame_id_to;      /* frames_kepts_ids_* */
  uint8_t transform_est_success;
  uint8_t pad[5];
  double  position[3];
  double  orientation[4];
  double  covariance[36];
} sf_separator;            /* 360 bytes */

typedef struct sf_match {  /* one element of DataHandler.find_matches' return value         */
  int32_t idx_local;
  int32_t idx_other;
  double  distance;        /* Euclidean distance of the pair (float64)                      */
} sf_match;

typedef struct sf_context* sf_handle;

/* ---- lifecycle ----------------------------------------------------------------------------- */
int  sf_abi_version(void);
void sf_default_params(sf_params* p);
/* device: HIP device ordinal.  Fails with SF_ENODEV when no GPU is visible (no CPU fallback). */
int  sf_create(const sf_params* p, int device, sf_handle* out);
void sf_destroy(sf_handle h);
const char* sf_last_error(sf_handle h);   /* h may be NULL: last create() error              */
int  sf_get_params(sf_handle h, sf_params* out);
/* Make the handle issue all work on a caller-owned hipStream_t (e.g. torch's current stream).  Call it before the first
   sf_step_issue: the step pipeline picks its further streams relative to this one, once (see sf_stream_placement in
   sf_experimental.h); a stream set later is used, but the other streams are not picked again. */
int  sf_set_stream(sf_handle h, void* hip_stream);
int  sf_synchronize(sf_handle h);

/* ---- NetVLAD nearest-neighbour stage (data_handler.py:166-209, 297-301, 402-408, 437-438) --- */
/* Append n descriptors of `dim` float64 values (FindMatches.srv:1 wire type) to the local
   (self.local_descriptors) or received (self.received_descriptors) database.                */
int  sf_nn_append_local(sf_handle h, const double* desc, int32_t n, int32_t dim);
int  sf_nn_append_received(sf_handle h, const double* desc, int32_t n, int32_t dim);
/* Same, from float32 rows already resident in device memory (bulk ingest; MI355X-native).   */
int  sf_nn_append_local_f32_device(sf_handle h, const float* d_desc, int32_t n, int32_t dim);
int  sf_nn_append_received_f32_device(sf_handle h, const float* d_desc, int32_t n, int32_t dim);
/* The same for IEEE binary16 descriptors (BASELINE configs[4] ships NetVLAD in fp16): n x dim halfs in device
   memory, converted exactly to the fp32 database rows.                                                    */
int  sf_nn_append_local_f16_device(sf_handle h, const uint16_t* d_desc, int32_t n, int32_t dim);
int  sf_nn_append_received_f16_device(sf_handle h, const uint16_t* d_desc, int32_t n, int32_t dim);
int  sf_nn_sizes(sf_handle h, int32_t* n_local, int32_t* n_received);
/* local_kf_already_used.append / other_kf_already_used.append / add_frames_kept_pairs_to_ignore
   These three change the state every query reads: while sf_step_issue has steps in flight they first wait for ALL of
   them on the host (the steps were issued on the masks as they were) -- a caller that marks after every accepted
   match serialises the step pipeline; mark between retiring a batch of steps and issuing the next.  Arguments are
   validated before that wait; a step whose re-run failed makes the call return that error (the mask is untouched). */
int  sf_nn_mark_local_used(sf_handle h, int32_t idx_local);
int  sf_nn_mark_other_used(sf_handle h, int32_t idx_other);
int  sf_nn_ignore_pair(sf_handle h, int32_t idx_local, int32_t idx_other);
int  sf_nn_reset(sf_handle h);
/* Switch sf_params.nn_precision (0 / 1, see above) on a live handle; databases are kept.        */
int  sf_nn_set_precision(sf_handle h, int32_t nn_precision);
/* DataHandler.find_matches(): writes up to `cap` matches, *n_out = number found.
   Returns SF_EINVAL when either database is empty (the reference guards this at
   data_handler.py:308).                                                                     */
int  sf_nn_find_matches(sf_handle h, sf_match* out, int32_t cap, int32_t* n_out);
/* Per-row minima of the last find_matches call (diagnostics / tests): n_local entries.      */
int  sf_nn_last_row_minima(sf_handle h, double* dist, int32_t* idx, int32_t cap);
/* The sequential tail of DataHandler.find_matches (data_handler.py:191-205: rows sorted by their minimum, the
   walk with its "idx_other already taken" and break rules) on caller-provided per-row minima -- host work only.
   For a node that shards the LOCAL rows of one query over its GPUs: every rank searches its
   block (sf_nn_row_minima_device below), the minima are all-gathered, and every rank runs this
   identical walk.  row_min: n_local float64 (+inf = no candidate under the threshold), row_arg: column of each. */
int  sf_nn_walk(sf_handle h, const double* row_min, const int32_t* row_arg, int32_t n_local, int32_t n_received,
                sf_match* out, int32_t cap, int32_t* n_out);
/* The NN kernels of this handle's local rows (data_handler.py:166-189 on ONE rank's block of rows) WITHOUT the walk
   and without a host round trip: d_row_min[n_local] (float64) and d_row_arg[n_local] (int32) are DEVICE buffers
   (typically a slice of the all-gather's send block), filled asynchronously on the handle's stream; rows with no
   column under netvlad_distance report (+inf, 0) on the filter path.  With nn_precision 1 the prefix filter runs at
   the ladder level the handle last settled on; d_status[0] (device int32) = 1 when the candidate set was denser than
   that level's sparse limit -- the minima are then undefined and the caller takes sf_nn_find_matches +
   sf_nn_last_row_minima (which walks the ladder) -- and 0 otherwise.                                           */
int  sf_nn_row_minima_device(sf_handle h, double* d_row_min, int32_t* d_row_arg, int32_t* d_status);
/* The walk of sf_nn_walk ON THE DEVICE (data_handler.py:191-205), asynchronous on the handle's stream, for minima that
   are already there: d_row_min / d_row_arg as sf_nn_row_minima_device leaves them (or the all-gathered minima of a node
   that shards the local rows), d_status (may be NULL) a device int32 that voids the walk (0 matches) when non-zero.
   Writes up to `cap` matches, walk order, to d_matches and their number to d_n_matches[0]; both may be device memory
   or host-pinned memory the device can write.  netvlad_distance and netvlad_max_matches_nb are the handle's.     */
int  sf_nn_walk_device(sf_handle h, const double* d_row_min, const int32_t* d_row_arg, const int32_t* d_status,
                       int32_t n_local, int32_t n_received, sf_match* d_matches, int32_t cap, int32_t* d_n_matches);
/* Descriptor dimensions the fp16 filter contracted in the last find_matches call (0: exact path). */
int  sf_nn_last_filter_dims(sf_handle h, int32_t* dims);

/* ---- device-resident keyframe feature store (data_handler.py:268 geometric_feats) ----------- */
int  sf_store_add_keyframe(sf_handle h, const sf_features* f, int32_t* out_slot);
/* Bulk ingest from device memory: n keyframes, each with exactly `rows` features of `cols`
   bytes; d_desc [n][rows][cols] u8, d_xyz [n][rows][3] f32, d_kp [n][rows] sf_keypoint.     */
int  sf_store_add_keyframes_device(sf_handle h, int32_t n, int32_t rows, int32_t cols,
                                   const uint8_t* d_desc, const float* d_xyz,
                                   const sf_keypoint* d_kp, int32_t* out_first_slot);
int  sf_store_size(sf_handle h, int32_t* n_slots);
int  sf_store_clear(sf_handle h);

/* ---- features of one stereo keyframe -------------------------------- */
/* replaces: RegistrationVis::getFeaturesImpl (myRegistrationVis.cpp:343-436: descriptors for the given keypoints,
   3D keypoints of the stereo pair, removal of keypoints without a finite 3D point) as called by
   StereoCamGeometricTools::getFeaturesAndDescriptor (stereoCamGeometricTools.cpp:100-120), writing the keyframe
   straight into the device-resident store.  Corner detection (sf_detect_corners_device) and the right-image
   position of every corner (sf_stereo_correspondences_device) are the two calls before this one.  */
typedef struct sf_stereo_camera {
  float fx, fy, cx, cy;        /* left camera (StereoCameraModel::left())                                  */
  float cx_right;              /* right camera cx (0: unknown, no principal-point correction)              */
  float baseline;              /* metres, > 0                                                              */
  float local_transform[12];   /* base <- camera optical frame, row-major 3x4 (CameraModel::localTransform) */
  float min_depth, max_depth;  /* Vis/MinDepth, Vis/MaxDepth; both 0 (the reference's defaults) keeps the
                                  keypoints whose 3D point is NaN                                          */
} sf_stereo_camera;
/* BRIEF test locations: [8 * bytes][4] int8 {x1, y1, x2, y2}, each within +-24 (the 48 px patch); bytes = 16, 32
   or 64.  A fresh handle holds a seeded Gaussian set (NOT OpenCV's table, which is not in the reference tree):
   install OpenCV's generated_<bytes>.i values here for descriptors identical to the reference build's.       */
int  sf_brief_set_pattern(sf_handle h, const int8_t* tests, int32_t bytes);
int  sf_brief_get_pattern(sf_handle h, int8_t* tests, int32_t cap_bytes, int32_t* bytes);
/* ---- NetVLAD descriptor inference --------------------------------- */
/* replaces: DataHandler.compute_descriptors (data_handler.py:143-164): `self.sess.run(self.net_out, ...)` of
   `nets.vgg16NetvladPca` (data_handler.py:63; netvlad_tf_open: VGG16 trunk to conv5_3, NetVLAD layer, WPCA), keeping
   the first n_out of the 4096 values like data_handler.py:157-158.  Weights come from the caller (the reference
   restores netvlad_tf_open's checkpoint, data_handler.py:70) in TensorFlow's own layouts:
     conv_kernel[i]   [3][3][Cin][Cout] (HWIO) of conv1_1, conv1_2, conv2_1, ... conv5_3 (13 layers), conv_bias[i] [Cout]
     average_rgb      [3]              assignment [512][clusters] (the 1 x 1 kernel, no bias)
     cluster_centers  [512][clusters]  wpca_kernel [512 * clusters][pca_dim], wpca_bias [pca_dim]                  */
typedef struct sf_netvlad_weights {
  const float* conv_kernel[13];
  const float* conv_bias[13];
  const float* average_rgb;
  const float* assignment;
  const float* cluster_centers;
  const float* wpca_kernel;
  const float* wpca_bias;
  int32_t      clusters;   /* 64 in the reference's network */
  int32_t      pca_dim;    /* 4096 */
} sf_netvlad_weights;
int  sf_netvlad_load(sf_handle h, const sf_netvlad_weights* w);      /* host pointers; copies and transposes */
/* d_image_rgb: [height][width][3] float32 on the device (the values the reference feeds its placeholder,
   data_handler.py:60-61); d_out: n_out floats = the first n_out values of the unit-norm descriptor, ready for
   sf_nn_append_local_f32_device.  Asynchronous on the handle's stream -- except the FIRST call at an image size,
   which measures every convolution layer in its tile / K-step configurations (about 40 ms, synchronous) and keeps
   the fastest for that size (neither changes the order of a pixel's sums: the descriptor's bits do not depend on it).                                                                                     */
int  sf_netvlad_infer_device(sf_handle h, const float* d_image_rgb, int32_t width, int32_t height, float* d_out,
                             int32_t n_out);
/* A batch, as DataHandler.compute_descriptors feeds the network (data_handler.py:149-156: up to netvlad_batch_size = 3
   queued images per call): n_images images of one size back to back in d_images_rgb, d_out [n_images][n_out].  Same
   bits per image as the single-image call; the WPCA matrix (537 MB) is read once per group of up to four images, and
   when the image height is a multiple of 16 the group's trunk runs as one vertical stack of its images (three times the
   workgroups in the late layers: 0.89 instead of 1.12 ms per 640 x 480 image at n_images = 3).  The first call with a
   given (size, stack height) measures its layer configurations like the single-image call does.                     */
int  sf_netvlad_infer_batch_device(sf_handle h, const float* d_images_rgb, int32_t n_images, int32_t width,
                                   int32_t height, float* d_out, int32_t n_out);

/* Corner detection of the reference's default feature type (rtabmap GFTT/BRIEF: Feature2D::generateKeypoints ->
   cv::goodFeaturesToTrack, called from myRegistrationVis.cpp:281-283), on the device: minimum-eigenvalue response
   (blockSize 3, Sobel 3), corners = local maxima above quality_level * max, strongest first, at least min_distance
   apart, at most max_corners (<= 0: no limit).  d_kpts_out receives up to `cap` keypoints {x, y, size 3, angle -1,
   response 0, octave 0, class_id -1} in that order; *n_out = corners found.  rtabmap's defaults: max_corners =
   Vis/MaxFeatures (1000), quality_level 0.001, min_distance 3.  Synchronises the stream twice (candidate count,
   result count).                                                                                              */
int  sf_detect_corners_device(sf_handle h, const uint8_t* d_image, int32_t width, int32_t height, int32_t pitch,
                              int32_t max_corners, double quality_level, double min_distance,
                              sf_keypoint* d_kpts_out, int32_t cap, int32_t* n_out);
/* Stereo correspondence of the corners: replaces Feature2D::generateKeypoints3D's call of
   StereoOpticalFlow::computeCorrespondences [upstream rtabmap] behind myRegistrationVis.cpp:382 --
   cv::calcOpticalFlowPyrLK(left, right, corners, winSize, maxLevel, {COUNT + EPS, iterations, epsilon},
   OPTFLOW_LK_GET_MIN_EIGENVALS, min_eig_threshold) followed by the disparity gate (status = 0 where left.x - right.x
   is <= min_disparity or > max_disparity).  sf_stereo_flow_defaults fills rtabmap's defaults: Stereo/WinWidth 15,
   Stereo/WinHeight 3, Stereo/MaxLevel 5 (3 before rtabmap 0.20), Stereo/Iterations 30, Stereo/Eps 0.01,
   Stereo/MinDisparity 0.5, Stereo/MaxDisparity 128, threshold 1e-4.                                              */
typedef struct sf_stereo_flow_params {
  int32_t win_width, win_height;    /* > 2 each, win_width * win_height <= 1024                      */
  int32_t max_level;                /* 0 .. 15; fewer levels are used when the image is small         */
  int32_t iterations;               /* clamped to 0 .. 100 like cv::TermCriteria in calcOpticalFlowPyrLK */
  double  epsilon;                  /* clamped to 0 .. 10, compared squared with |delta|^2            */
  float   min_disparity, max_disparity;
  float   min_eig_threshold;
} sf_stereo_flow_params;
void sf_stereo_flow_defaults(sf_stereo_flow_params* p);
/* d_left / d_right: rectified 8-bit images on the device (height rows of `pitch` bytes); d_kpts: n corners of the
   left image (sf_detect_corners_device's output).  Outputs on the device, asynchronous on the handle's stream:
   d_right_xy [n][2] (cv::Point2f of every corner in the right image), d_status [n], d_right_x [n] (optional: the x
   alone, the layout sf_extract_keyframe_device takes), d_err [n] (optional: the minimum eigenvalue of the level-0
   window, what calcOpticalFlowPyrLK returns as err under OPTFLOW_LK_GET_MIN_EIGENVALS).  params NULL = defaults.  */
int  sf_stereo_correspondences_device(sf_handle h, const uint8_t* d_left, const uint8_t* d_right, int32_t width,
                                      int32_t height, int32_t pitch, const sf_keypoint* d_kpts, int32_t n,
                                      const sf_stereo_flow_params* params, float* d_right_xy, uint8_t* d_status,
                                      float* d_right_x, float* d_err);
/* d_left: 8-bit image on the device (height rows of `pitch` bytes); d_kpts: n corners; d_right_x: their x in the
   right image (NULL: no 3D); d_status: per-corner validity of d_right_x (NULL: all valid).  Appends ONE keyframe
   to the store; *out_slot = its slot, *out_rows = features kept (the call synchronises the stream to read it;
   pass NULL to stay asynchronous).  Optional copies for the wire (GetFeatsAndDesc response), device pointers
   sized for n rows, any may be NULL: d_desc_out [n][bytes], d_xyz_out [n][3], d_kpts_out [n].               */
int  sf_extract_keyframe_device(sf_handle h, const uint8_t* d_left, int32_t width, int32_t height, int32_t pitch,
                                const sf_keypoint* d_kpts, const float* d_right_x, const uint8_t* d_status,
                                int32_t n, const sf_stereo_camera* cam, int32_t* out_slot, int32_t* out_rows,
                                uint8_t* d_desc_out, float* d_xyz_out, sf_keypoint* d_kpts_out);
/* The GetFeatsAndDesc service handler in ONE call on host buffers -- replaces the body of
   StereoCamGeometricTools::getFeaturesAndDescriptor (stereoCamGeometricTools.cpp:100-120: SensorData(img_l, img_r, cam_),
   registrationPipeline_->getFeatures(...) = myRegistrationVis.cpp:190-439, then the three ...ToROS conversions):
   sf_detect_corners_device -> sf_stereo_correspondences_device -> sf_extract_keyframe_device on device copies of the
   pair.  left / right: rectified MONO8 images in host memory (what cv_bridge::toCvCopy returns at :104-105), `pitch`
   bytes per row.  det NULL = rtabmap's defaults (Vis/MaxFeatures 1000, GFTT/QualityLevel 0.001, GFTT/MinDistance 3),
   flow NULL = its Stereo/ defaults.  Outputs in host memory, sized for cap_rows rows (any may be NULL): desc_out
   [rows][bytes of the BRIEF table], xyz_out [rows][3], kpts_out [rows]; *rows_out = features of the keyframe (may
   exceed cap_rows: only cap_rows are copied); *slot_out (optional) = its slot in the device-resident store, what
   sf_verify_pairs refers to later.  Synchronous.                                                              */
typedef struct sf_detector_params {
  int32_t max_features;      /* 1 .. 32767 (KeyPointVec.size is an int16) */
  double  quality_level;
  double  min_distance;
} sf_detector_params;
void sf_detector_defaults(sf_detector_params* p);
int  sf_get_features_and_descriptor(sf_handle h, const uint8_t* left, const uint8_t* right, int32_t width, int32_t height,
                                    int32_t pitch, const sf_stereo_camera* cam, const sf_detector_params* det,
                                    const sf_stereo_flow_params* flow, uint8_t* desc_out, float* xyz_out,
                                    sf_keypoint* kpts_out, int32_t cap_rows, int32_t* rows_out, int32_t* slot_out);

/* A batch of keyframes: n_keyframes rectified MONO8 pairs of one size already in DEVICE memory
   (pair i at d_left / d_right + i * image_stride bytes), through the same three stages as sf_get_features_and_descriptor
   in ONE launch sequence -- every stage runs once over the whole batch, the corner counts stay in device memory between
   the stages, and the host is never waited for (asynchronous on the handle's stream).  The keyframes land in the store
   slots *first_slot_out .. + n_keyframes - 1; per keyframe the results are the single call's, byte for byte.  Optional
   device outputs, each sized for n_keyframes x det->max_features rows (keyframe i at row i * max_features; any may be
   NULL): d_rows_out [n_keyframes] features kept, d_desc_out, d_xyz_out, d_kpts_out as in the single call.  Images up to
   about 1.2 Mpixel (the corner selection keeps its bitmap in LDS).                                                 */
int  sf_get_features_and_descriptor_batch_device(sf_handle h, const uint8_t* d_left, const uint8_t* d_right,
                                                 int32_t n_keyframes, int32_t width, int32_t height, int32_t pitch,
                                                 size_t image_stride, const sf_stereo_camera* cam,
                                                 const sf_detector_params* det, const sf_stereo_flow_params* flow,
                                                 int32_t* first_slot_out, int32_t* d_rows_out, uint8_t* d_desc_out,
                                                 float* d_xyz_out, sf_keypoint* d_kpts_out);

/* ---- geometric verification (stereoCamGeometricTools.cpp:122-178) ---------------------------- */
/* One estimate_transformation service call on host buffers.                                  */
int  sf_estimate_transform(sf_handle h, const sf_features* from, const sf_features* to,
                           sf_result* out);
/* n service calls in one launch sequence, host buffers (features are staged into scratch
   store slots).                                                                              */
int  sf_estimate_transform_batch(sf_handle h, const sf_features* from, const sf_features* to,
                                 int32_t n, sf_result* out);
/* n candidate pairs given as store slots (from_slot[i] -> to_slot[i]); results to host.      */
int  sf_verify_pairs(sf_handle h, const int32_t* from_slot, const int32_t* to_slot, int32_t n,
                     sf_result* out);
/* Fully device-resident variant: slot arrays and results live in device memory (e.g. a torch
   tensor's data_ptr()); asynchronous on the handle's stream.                                 */
int  sf_verify_pairs_device(sf_handle h, const int32_t* d_from_slot, const int32_t* d_to_slot,
                            int32_t n, sf_result* d_out);
/* Verify the candidates an NN query returned: candidate i is the pair (store slot slot_base_other +
   matches[i].idx_other  ->  slot slot_base_local + matches[i].idx_local), i.e. the (frame of the
   querying robot, frame of the computing robot) pair find_separators.py:85-91 sends to
   estimate_transformation.  `matches` is HOST memory (the output of sf_nn_find_matches), d_out device
   memory for n records.  Asynchronous on the handle's stream; `matches` may be reused on return.   */
int  sf_verify_matches_device(sf_handle h, const sf_match* matches, int32_t n, int32_t slot_base_other,
                              int32_t slot_base_local, sf_result* d_out);
/* Compact the ACCEPTED results (success != 0) of a verification, in candidate order, into d_accepted
   (device, room for n records) and write every candidate's success flag to d_flags (device, n bytes,
   may be NULL): what find_separators.py:97-131 forwards as separators, resp. feeds back as failures.
   Returns the number of accepted records in *n_accepted (host); synchronises the stream.           */
int  sf_compact_accepted_device(sf_handle h, const sf_result* d_results, int32_t n, sf_result* d_accepted,
                                uint8_t* d_flags, int32_t* n_accepted);
/* The same compaction without the synchronisation: the count is left in device memory at d_n_accepted
   (int32), everything asynchronous on the handle's stream -- for callers that ship count, flags and a
   speculative prefix of the accepted records to the host (or stamp the count into an exchange buffer) behind
   ONE synchronisation of their own.                                                                    */
int  sf_compact_accepted_device_async(sf_handle h, const sf_result* d_results, int32_t n, sf_result* d_accepted,
                                      uint8_t* d_flags, int32_t* d_n_accepted);
/* Correspondences found by the two matching passes of the LAST verify call for pair `i`
   (tests / diagnostics): pairs (from_feature, to_feature), ascending from_feature.  Calls with more
   than 131072 pairs are processed in chunks; `pair` then indexes the LAST chunk.  Needs SF_OPT_DEBUG_CORR
   set BEFORE that verify call (SF_EINVAL otherwise) when the fused kernel ran it.                  */
int  sf_debug_correspondences(sf_handle h, int32_t pair, int32_t pass, uint16_t* from_idx,
                              uint16_t* to_idx, int32_t cap, int32_t* n_out);

/* ---- separator records ------------------------------------------------------------------------ */
/* Pack accepted/failed results into ReceiveSeparators rows (host side, no GPU work).          */
int  sf_pack_separators(const sf_result* res, int32_t n, int8_t robot_from, int8_t robot_to,
                        const int16_t* kf_from, const int16_t* kf_to, const int16_t* frame_from,
                        const int16_t* frame_to, sf_separator* out);

/* ---- multi-GPU exchange (RCCL over xGMI) ------------------------------------------------------- */
/* The path shards with no collective in the compute (pairs are independent); the one exchange is an
   all-gather of accepted separator records, what the reference moves between robots as
   ReceiveSeparators requests (communication.cpp:15-52).  One handle (one GPU) per rank.          */
#define SF_COMM_ID_BYTES 128
int  sf_comm_unique_id(uint8_t* out, int32_t cap);          /* rank 0 creates, the host distributes */
int  sf_comm_init(sf_handle h, const uint8_t* unique_id, int32_t rank, int32_t world);
int  sf_comm_destroy(sf_handle h);
/* d_local: n_local records in device memory.  d_all: world * cap_per_rank records in device memory,
   rank r's records at d_all + r * cap_per_rank; counts (host): world entries.  Synchronous; one collective (the
   count travels in a header slot of the block).                                                       */
int  sf_allgather_separators(sf_handle h, const sf_separator* d_local, int32_t n_local,
                             sf_separator* d_all, int32_t cap_per_rank, int32_t* counts);
/* The same exchange with no host in it: d_send = cap_per_rank + 1 record slots, slot 0 a header whose first int32
   is this rank's count STAMPED ON THE DEVICE (e.g. by sf_compact_accepted_device_async writing its count there),
   slots 1.. the records; d_all = world such blocks (block r = rank r).  ONE collective, asynchronous on the handle's
   stream; the caller reads the counts out of the gathered headers behind its own synchronisation.            */
/* The same collective on an opaque block of bytes_per_rank bytes (device memory, asynchronous on the handle's
   stream): the row minima of the row-sharded NN stage travel this way.  d_all: world * bytes_per_rank bytes.   */
int  sf_allgather_bytes_device(sf_handle h, const void* d_send, void* d_all, size_t bytes_per_rank);
int  sf_allgather_separators_device(sf_handle h, const sf_separator* d_send, sf_separator* d_all,
                                    int32_t cap_per_rank);

/* sf_nn_find_matches followed by sf_verify_matches_device of what it returned, as ONE call (the loop body of
   find_separators.py:59-133 when both robots' keyframes live in this handle): `out` / `*n_out` as
   sf_nn_find_matches, d_out[i] (device, caller-allocated, >= cap records) = result of match i, asynchronous
   on the handle's stream.  Same outputs as the two calls; when the walk may return every local row
   (cap and netvlad_max_matches_nb >= local rows, nn_precision = 1) the candidates of the NN filter are
   verified SPECULATIVELY on the device while the host still reduces / sorts / walks them, which takes the
   host part of the NN stage off the critical path.                                                     */
int  sf_find_matches_and_verify_device(sf_handle h, int32_t slot_base_other, int32_t slot_base_local,
                                       sf_match* out, int32_t cap, int32_t* n_out, sf_result* d_out);
/* d_out may be NULL: the results then stay in the handle's own block (sf_step_* and sf_experimental.h consume them
   there without a gathered copy of all records).                                                                  */

/* ---- the caller's loop body as a begin / retire pair ------------------------------------------------------------ */
/* replaces: one iteration of find_separators.py:59-133 when both robots' keyframes live in this handle --
     sf_step_issue   = s_find_matches_query (:63) + one s_ans_est_transform per returned candidate (:83-95), QUEUED: the
                       NN kernels, the argsort + walk of DataHandler.find_matches (data_handler.py:187-205) ON THE DEVICE
                       and the verification of the walk's matches are all queued on one stream and the call returns
                       without waiting for any of it (SF_OPT_STEP_DEVICE_WALK);
     sf_step_retire  = what the loop then does with the outcomes (:97-133): for every match, in the order
                       DataHandler.find_matches returned them, whether the estimation succeeded and, if so, its
                       PoseWithCovariance -- the rows of the ReceiveSeparators request (sf_pack_separators packs them).
                       It is the only call of the pair that waits for the device.
   Up to SF_OPT_STEP_DEPTH steps (default 6) may be in flight, so the device never waits for the host (bench.py,
   examples/bench_cli.cpp):  issue(0) .. issue(D-1); then retire(k), issue(k + D) ...; retire the rest.
   The steps in flight are dealt over SF_OPT_STEP_LANES streams (SF_OPT_STEP_OVERLAP): work the caller queued on the
   handle's stream before sf_step_issue is waited for, and every call that changes a database or a mask
   (sf_store_*, sf_nn_append_*, sf_nn_mark_*, sf_nn_ignore_pair, sf_nn_reset) first waits for the steps in flight:
   a step sees the databases and masks as they were when it was issued.
   Accepted results leave the verification kernel for host-pinned memory the moment they are final (no compaction
   launch, no copy); launch shapes the stream does not cover (stage kernels, more than 131 072 matches) end with an
   ordered compaction inside the same two calls.  With nn_precision 1 the NN filter runs at the prefix level the
   handle last settled on; a candidate set too dense for that level is detected on the device, and sf_step_retire then
   runs that one query again through the prefix ladder (sf_nn_find_matches' path) before it returns -- results are
   identical either way; only with a mirror set (below) such a step returns SF_ERANGE, since the mirror missed it.
   Pointers in sf_step_result are owned by the handle and stay valid until the next sf_step_retire.               */
typedef struct sf_step_result {
  const sf_match*  matches;          /* n_matches candidates, walk order (data_handler.py:191-205)                   */
  const int32_t*   record_of_match;  /* n_matches: index into `records` of the match's result, -1 = estimation failed
                                        (transform_est_success = 0: find_separators.py:119-126 feeds the ignore list) */
  const sf_result* records;          /* accepted results (success = 1) in host-pinned memory; completion order when
                                        streamed, match order otherwise: always go through record_of_match           */
  int32_t          n_matches;
  int32_t          n_records;        /* accepted results present (>= n_accepted: a local row with two candidates under
                                        the threshold had both verified, the walk kept one)                          */
  int32_t          n_accepted;       /* matches whose estimation succeeded                                           */
  int32_t          streamed;         /* 1: the records streamed out of the kernel, 0: compacted behind it            */
  const sf_result* d_records;        /* the same n_records records, same order, in DEVICE memory (NULL while a mirror is
                                        set): what a multi-GPU host hands to its all-gather after the retire -- by a
                                        device-to-device copy into its send buffer (sf_memcpy_device_async) or directly;
                                        valid until the next sf_step_retire like the other pointers                     */
} sf_step_result;
int  sf_step_issue(sf_handle h, int32_t slot_base_other, int32_t slot_base_local);
int  sf_step_retire(sf_handle h, sf_step_result* out);      /* the OLDEST step in flight; waits for its verification */
/* bytes from device memory to device memory, asynchronous on `hip_stream` (NULL: the handle's stream) -- e.g. a retired
   step's sf_step_result.d_records -> the send buffer of the caller's collective, on the stream the caller runs its
   collectives from (a stream of its own keeps that traffic out of the way of the steps in flight).  A copy out of a
   step's d_records is remembered: the step that next uses that block waits for it before it writes the buffer again,
   whatever stream the copy was queued on. */
int  sf_memcpy_device_async(sf_handle h, void* d_dst, const void* d_src, size_t bytes, void* hip_stream);
/* Optional second destination of every accepted record, on the device -- e.g. the send buffer of the all-gather that
   follows (sf_allgather_separators_device: slot 0 = header, records from slot 1): d_records2[slot] receives the record
   the host block receives at the same slot and *d_counter (a device word the CALLER zeroes before each sf_step_issue,
   e.g. the count in that header) counts the slots taken.  cap = record slots behind d_records2; a cap below the number
   of verified candidates of a query (local rows * 9 / 8 + 256) switches that query to the compaction (nothing is ever
   dropped).  NULL, NULL, 0 removes the mirror.  Not while a step is in flight.
   While a mirror is set sf_step_issue admits as many steps in flight as the mirror has buffers (1 here, 2 with
   sf_step_mirror_pair) and fails with SF_EINVAL beyond -- step k + 2 would zero and overwrite what step k wrote:
   the loop is issue, [issue,] retire + hand the buffer to the collective, issue, ... (ABI 6).                       */
int  sf_step_mirror(sf_handle h, sf_result* d_records2, uint32_t* d_counter, int32_t cap);
/* The same with TWO destinations that alternate with the steps (the first sf_step_issue after this call writes the
   even pair, the next one the odd pair, ...): with two send buffers the all-gather of step k runs beside the
   verification of step k + 1 instead of in front of it -- the caller orders only the REUSE of a buffer (step k + 2)
   behind the collective that read it.  Both pairs NULL removes the mirror.  Not while a step is in flight.        */
int  sf_step_mirror_pair(sf_handle h, sf_result* d_records_even, uint32_t* d_counter_even,
                         sf_result* d_records_odd, uint32_t* d_counter_odd, int32_t cap);
/* After sf_step_mirror_pair: lets the odd steps run on the handle's second stream as they do without a mirror
   (SF_OPT_STEP_OVERLAP) and returns the two hipStream_t.  From here on the caller orders everything it does with the
   even pair -- zeroing the counter, the collective that reads the buffer -- on *stream_even and everything with the
   odd pair on *stream_odd.  With SF_OPT_STEP_OVERLAP off both are the handle's stream.  Ends with the next
   sf_step_mirror / sf_step_mirror_pair call.  Not while a step is in flight.                                     */
int  sf_step_mirror_streams(sf_handle h, void** stream_even, void** stream_odd);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Kernel ids for sf_prof_get */
enum {
  SF_K_MATCH = 0,      /* global Hamming kNN-2 + NNDR + uniqueness (pass 1)                   */
  SF_K_RANSAC1 = 1,    /* RANSAC 3D-3D + refine, pass 1                                       */
  SF_K_GUIDED = 2,     /* guess-guided window matching (pass 2)                               */
  SF_K_RANSAC2 = 3,    /* RANSAC 3D-3D + refine, pass 2                                       */
  SF_K_NN = 4,         /* NetVLAD distance matrix + fused row arg-min                         */
  SF_K_NN_SELECT = 5,  /* f64 re-evaluation of the row minima                                 */
  SF_K_NN_FILTER = 6,  /* fp16 MFMA candidate filter (nn_precision = 1)                       */
  SF_K_NN_REFINE = 7,  /* exact f64 distance of every filter survivor                         */
  SF_K_FUSED = 8,      /* fused per-pair pipeline: match + RANSAC + guided + RANSAC + result    */
  SF_K_NN_WALK = 9,    /* argsort of the row minima + the walk (data_handler.py:191-205), on the device */
  SF_K_BA = 10,        /* two-view bundle adjustment of a pass's estimate (a launch of its own since ABI 6) */
  SF_K_COUNT = 11
};
/* When enabled every kernel launch is bracketed by hipEvents on the handle's stream.          */
int  sf_prof_enable(sf_handle h, int on);
int  sf_prof_reset(sf_handle h);
int  sf_prof_get(sf_handle h, int kernel, int64_t* launches, double* total_ms);
const char* sf_kernel_name(int kernel);

/* Execution options of a live handle (none changes any output byte; the same switches are read from the
   environment at sf_create: SF_MATCH_MFMA, SF_FUSED, SF_OVERLAP).  Returns SF_EINVAL for an unknown option. */
enum {
  SF_OPT_MATCH_MFMA = 0,  /* 1 (default): Hamming table on the fp4 matrix cores; 0: xor + popcount on the VALU   */
  SF_OPT_FUSED = 1,       /* 1 (default): one fused launch per chunk (3D-3D estimator); 0: the stage kernels     */
  SF_OPT_OVERLAP = 2,     /* 1: batches >= 4096 pairs as two halves on two streams; 0 (default): one stream     */
  SF_OPT_CHAIN_WAVES = 3, /* round 1's narrower motion-estimation chains (measured slower, removed): accepted, no effect */
  SF_OPT_NN_FULL_FILTER = 5, /* 1: the fp16 NN filter (nn_precision = 1) always contracts the FULL descriptor length
                             instead of climbing its adaptive prefix ladder (128 / 512 / full): the cost of a data set
                             whose prefixes are uninformative; matches are identical either way                        */
  SF_OPT_STEP_OVERLAP = 6, /* 1 (default): the two steps sf_step_issue keeps in flight run on two streams (the odd ones on a
                              stream and workspace of the handle's own), so that the emptying tail of one step's
                              verification overlaps the NN stage and the first workgroups of the next: +9-12 % steps per
                              second (3D-3D), +14 % (PnP), results unchanged.  0: every step on the handle's stream.
                              Ignored while sf_step_mirror is set (the caller's collective is ordered on the handle's
                              stream) unless sf_step_mirror_streams handed the two streams to the caller.  Either way a step's results are complete when sf_step_retire returns.            */
  SF_OPT_STEP_SPLIT = 7,  /* 1: while the steps are dealt over several streams (SF_OPT_STEP_OVERLAP) the 3D-3D verification
                             of a step runs as one matching launch over all candidates + one chain launch over the
                             survivors instead of the fused kernel (256-bit descriptors, K <= 512 features, 2 048 .. 65 536
                             candidates -- other shapes keep the fused kernel).  Default 1 again since round 5: the matching
                             launch's scan is software-pipelined inside a wavefront and the form is the faster one by
                             1-1.5 % (23.0 against 22.7 M pairs/s, profiles/r05u_*; round 4, before that: the fused
                             kernel 22.0 against 20.3).  0: always the fused kernel.  Results unchanged.              */
  SF_OPT_STEP_DEPTH = 8,  /* steps sf_step_issue keeps in flight, 1 .. 16 (default 6).  Not while a step is in flight.  */
  SF_OPT_STEP_LANES = 9,  /* streams the steps in flight are dealt over, 1 .. 4 (default 3; step k runs on stream
                             k mod lanes; with a mirror set at most 2).  Not while a step is in flight.              */
  SF_OPT_STEP_DEVICE_WALK = 10, /* 1 (default): sf_step_issue never waits for the device -- the argsort + walk of
                             data_handler.py:191-205 run on the device between the NN kernels and the verification;
                             0: round 3's form (the call returns once the host has walked the row minima).  Results
                             unchanged.  Not while a step is in flight.                                              */
  SF_OPT_STEP_SPECULATE = 11, /* 1 (default): a device-resident step whose walk may return every local row
                             (netvlad_max_matches_nb >= local rows, nn_precision 1) queues the verification of EVERY NN
                             filter candidate straight behind the filter and runs the exact re-evaluation, the row
                             minima and the walk on a second stream beside it; 0: always NN -> walk -> verification of
                             the walk's matches on one stream.  Results unchanged.  Not while a step is in flight.   */
  SF_OPT_DEBUG_CORR = 4   /* 1: the fused kernel also copies every pair's correspondence lists, headers and pass states
                             to the global workspace, which sf_debug_correspondences reads (default 0: they never
                             leave the workgroup's LDS; the stage kernels always keep them in the workspace)       */
};
int  sf_set_option(sf_handle h, int32_t option, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* SEPFINDER_H */
