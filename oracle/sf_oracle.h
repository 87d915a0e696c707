/*
 * sf_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's separator-finder hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (multi_robot_slam_separators_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - NN stage (sfo_find_matches): PINNED against scipy.spatial.distance.cdist + numpy.argsort
 *     driven through the statement sequence of data_handler.py:170-205 (tests/golden/nn_*.npz,
 *     generator oracle/gen_golden.py).
 *   - Matching / RANSAC / PnP stage: PARITY UNPINNED.  The reference delegates that arithmetic to
 *     rtabmap (unpinned master), PCL >= 1.7, OpenCV and FLANN, none of which are vendored under
 *     /root/reference or installed here, and the reference ships no tests or golden vectors.
 *     The restatement follows the reference's call sites line by line and the published
 *     algorithms of those libraries (cited per function); it is validated against planted
 *     ground truth and an independent numpy restatement (tests/), not against reference output.
 *
 * Citations: PKG = /root/reference/ros_ws/src/multi_robot_separators.
 */
#ifndef SF_ORACLE_H
#define SF_ORACLE_H

#include <stdint.h>
#include "../include/sepfinder.h"   /* POD wire structs + sf_params only */

#ifdef __cplusplus
extern "C" {
#endif

/* DataHandler.find_matches -- PKG/scripts/data_handler.py:166-209.
 * local [n_l][dim], received [n_r][dim] float64.  Masks: rows local_used[], columns
 * other_used[], pairs ignored[2*i]=(local,other).  Writes up to cap matches in walk order.
 * row_min / row_arg (optional, n_l entries) receive the masked per-row minimum and arg-min. */
int sfo_find_matches(const double* local, int n_l, const double* received, int n_r, int dim,
                     const int32_t* local_used, int n_local_used,
                     const int32_t* other_used, int n_other_used,
                     const int32_t* ignored_pairs, int n_ignored,
                     double netvlad_distance, int max_matches_nb,
                     sf_match* out, int cap, int* n_out,
                     double* row_min, int32_t* row_arg);

/* Global matching branch -- PKG/src/myRegistrationVis.cpp:826-895 with VWDictionary in
 * brute-force mode [upstream rtabmap VWDictionary::addNewWords].  Outputs the id-aligned
 * correspondences (ascending "from" index) and the word counts the 3D-3D gate uses. */
int sfo_match_global(const uint8_t* desc_from, int k_from, const uint8_t* desc_to, int k_to,
                     int cols, float nndr, int has3d_from, int has3d_to,
                     uint16_t* corr_from, uint16_t* corr_to, int* n_corr,
                     int* n_words_from, int* n_words_to, int* n_words_to_2d, int desc_type /* 0 = binary rows (Hamming), 1 = float32 rows (squared L2) */);

/* Guess-guided matching branch -- PKG/src/myRegistrationVis.cpp:476-825 (default sub-branch
 * :667-818, _guessMatchToProjection=false), exact radius search instead of FLANN's
 * approximate kd-tree.  guess = row-major 3x4 float (p_from = guess * p_to).
 * Returns *all_outside = 1 when no projected point falls in the image (:820-823). */
int sfo_match_guided(const sf_params* p, const float* guess,
                     const uint8_t* desc_from, const float* xyz_from, const sf_keypoint* kp_from,
                     int k_from,
                     const uint8_t* desc_to, const sf_keypoint* kp_to, int k_to, int has3d_to,
                     int cols,
                     uint16_t* corr_from, uint16_t* corr_to, int* n_corr,
                     int* n_words_from, int* n_words_to, int* n_words_to_2d, int* all_outside);

/* util3d::estimateMotion3DTo3D as called at PKG/src/myRegistrationVis.cpp:1122-1131
 * [upstream rtabmap util3d_motion_estimation.cpp / util3d_registration.cpp
 *  transformFromXYZCorrespondences; PCL RandomSampleConsensus +
 *  SampleConsensusModelRegistration].
 * Inputs are the id-aligned correspondences.  transform (row-major 3x4 float, p_from = T p_to)
 * is all-zero when null.  variance_out multiplies I6 (1.0 when untouched). */
typedef struct sfo_motion {
  float  transform[12];
  int    is_null;
  double variance;      /* covariance = variance * I6 (3D-3D); linear block (PnP)  */
  double variance_ang;  /* angular block of the covariance (PnP); = variance for 3D-3D */
  int    matches;
  int    inliers;
  int    ransac_best_iteration;   /* diagnostics */
  int    ransac_iterations_run;
  int    ransac_best_count;
  int    refine_rounds;
} sfo_motion;

/* Two-view bundle adjustment of one pass (myRegistrationVis.cpp:1192-1370; sf_oracle_ba.c).  mask: the motion
 * estimate's inliers over the pass's correspondences.  T in/out (p_from = T p_to); *n_inliers in/out; *is_null out. */
int sfo_bundle_adjust(const sf_params* p, const float* xyz_from, const sf_keypoint* kp_from, const float* xyz_to,
                      const sf_keypoint* kp_to, const uint16_t* corr_from, const uint16_t* corr_to,
                      const uint8_t* mask, int n_corr, float T[12], int* n_inliers, int* is_null, uint8_t* mask_out);

int sfo_estimate_motion_3d3d(const sf_params* p,
                             const float* xyz_from, const float* xyz_to,
                             const uint16_t* corr_from, const uint16_t* corr_to, int n_corr,
                             sfo_motion* out, uint8_t* inlier_mask /* optional, n_corr */);

/* util3d::estimateMotion3DTo2D as called at PKG/src/myRegistrationVis.cpp:1077-1091
 * [upstream rtabmap util3d_motion_estimation.cpp; cv::solvePnPRansac].  Restated in
 * sf_oracle_pnp.c (see its header for what is and is not OpenCV's arithmetic).
 * xyz_to may be NULL ("to" frame without 3D points). */
int sfo_estimate_motion_3d2d(const sf_params* p,
                             const float* xyz_from, const sf_keypoint* kp_to, const float* xyz_to,
                             const uint16_t* corr_from, const uint16_t* corr_to, int n_corr,
                             sfo_motion* out, uint8_t* inlier_mask /* optional, n_corr */);

/* estimateTransformation service -- PKG/src/stereoCamGeometricTools.cpp:122-178 driving
 * PKG/src/myRegistration.cpp:225-303 and PKG/src/myRegistrationVis.cpp:441-1410 twice. */
int sfo_estimate_transform(const sf_params* p, const sf_features* from, const sf_features* to,
                           sf_result* out);
/* Same with the intermediate correspondences exposed (tests). pass = 1 or 2. */
int sfo_estimate_transform_dbg(const sf_params* p, const sf_features* from, const sf_features* to,
                               sf_result* out,
                               uint16_t* c1_from, uint16_t* c1_to, int* n_c1,
                               uint16_t* c2_from, uint16_t* c2_to, int* n_c2);

/* Batch helper for the timed CPU baseline: n independent calls, optional OpenMP over pairs. */
int sfo_estimate_transform_batch(const sf_params* p, const sf_features* from,
                                 const sf_features* to, int n, sf_result* out, int threads);

/* Building blocks exposed for unit tests */
void   sfo_fit_rigid(const double* src, const double* dst, int n, double R[9], double t[3]);
double sfo_canon_log(double x);
void sfo_to3dof(float* T);
void sfo_interpolate_half(const float* A, const float* B, float* out);
void   sfo_sample_triplet(uint64_t seed, uint32_t iteration, uint32_t attempt, uint32_t m,
                          uint32_t out[3]);
int    sfo_num_threads(void);
void   sfo_set_num_threads(int n);   /* size of the OpenMP team of the batch / NN helpers */
void   sfo_sample_quad(uint64_t seed, uint32_t iteration, uint32_t attempt, uint32_t m, uint32_t out[4]);
int    sfo_quartic_roots(const double c[5], double r[4]);
int    sfo_p3p(const double P[3][3], const double f[3][3], double R[4][9], double t[4][3]);
double sfo_canon_atan2(double y, double x);

/* SURVEY section 8 row f3 (sf_oracle_extract.c): features of one stereo keyframe for given corners.  tests:
 * [8 * bytes][4] int8 {x1, y1, x2, y2}; outputs sized for n rows; *rows_out = rows kept.                          */
int sfo_extract_keyframe(const uint8_t* image, int32_t width, int32_t height, int32_t pitch, const sf_keypoint* kpts,
                         const float* right_x, const uint8_t* status, int32_t n, const sf_stereo_camera* cam,
                         const int8_t* tests, int32_t bytes, uint8_t* desc_out, float* xyz_out, sf_keypoint* kp_out,
                         int32_t* rows_out);

/* SURVEY section 8 row f3, the detector (sf_oracle_gftt.c): cv::goodFeaturesToTrack restated; *n_out = corners found
 * (only the first `cap` are written); eig_out (optional, width * height floats) receives the response map.      */
int sfo_detect_corners(const uint8_t* image, int32_t width, int32_t height, int32_t pitch, int32_t max_corners,
                       double quality_level, double min_distance, sf_keypoint* kpts_out, int32_t cap, int32_t* n_out,
                       float* eig_out);

/* SURVEY section 8 row f3, the stereo correspondence (sf_oracle_lk.c): cv::calcOpticalFlowPyrLK + rtabmap's disparity
 * gate restated.  right_xy [n][2], status [n], err [n] (optional), *levels_used (optional) = pyramid levels built. */
int sfo_stereo_correspondences(const uint8_t* left, const uint8_t* right, int32_t width, int32_t height, int32_t pitch,
                               const sf_keypoint* kpts, int32_t n, const sf_stereo_flow_params* prm, float* right_xy,
                               uint8_t* status, float* err, int32_t* levels_used);
void sfo_pyr_down(const uint8_t* src, int32_t w, int32_t h, int32_t pitch, uint8_t* dst);        /* cv::pyrDown, 8-bit */
void sfo_scharr_deriv(const uint8_t* src, int32_t w, int32_t h, int32_t pitch, int16_t* dst);    /* calcSharrDeriv */

#ifdef __cplusplus
}
#endif
#endif
