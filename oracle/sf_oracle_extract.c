/* sf_oracle_extract.c -- CPU ORACLE (test infrastructure, never linked into the product) for SURVEY.md section 8
 * row f3: the features of one stereo keyframe, as RegistrationVis::getFeaturesImpl produces them for
 * StereoCamGeometricTools::getFeaturesAndDescriptor (stereoCamGeometricTools.cpp:100-120):
 *   myRegistrationVis.cpp:343-354   descriptors for the given keypoints (detector->generateDescriptors)
 *   myRegistrationVis.cpp:356-383   3D keypoints of the stereo pair (detector->generateKeypoints3D)
 *   myRegistrationVis.cpp:384-425   removal of keypoints without a finite 3D point when a depth range is set
 *
 * PARITY UNPINNED: generateDescriptors / generateKeypoints3D live in rtabmap (Features2d.cpp, util3d_features.cpp,
 * util3d.cpp; pinned by the reference's Dockerfile to rtabmap 0.19 / OpenCV 3.x with opencv_contrib), none of which is
 * vendored in /root/reference or installed here.  What is restated below is their published algorithm:
 *   - BRIEF (opencv_contrib xfeatures2d brief.cpp): keypoints closer than PATCH_SIZE/2 + KERNEL_SIZE/2 = 28 px to
 *     the border are dropped; every bit compares two 9x9 box sums of the image taken from its integral image at
 *     offsets (x, y) from the rounded keypoint position; bit t of byte b = test 8 b + t, most significant first.
 *     The 8*bytes test locations are DATA here (OpenCV's table, generated_32.i, is not in the reference tree): the
 *     caller supplies them, and an integrator who wants the reference build's exact descriptors installs OpenCV's.
 *   - stereo 3D (util3d::generateKeypoints3DStereo + projectDisparityTo3D): disparity = x_left - x_right,
 *     W = baseline / (disparity + (cx_right - cx_left)), point = ((x - cx) W, (y - cy) W, fx W), kept when finite
 *     and inside (min_depth, max_depth], then moved to the base frame by the camera's local transform; anything
 *     else is (NaN, NaN, NaN).  The right-image position of each corner (rtabmap: pyramidal LK flow,
 *     Stereo::computeCorrespondences) is an INPUT.
 * Arithmetic: float, in the operation order written here, no contraction (-ffp-contract=off); the GPU kernels
 * (csrc/k_extract.hip) use the same order, so outputs are compared byte for byte.                                  */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sf_oracle.h"

#define SFO_BRIEF_PATCH 48
#define SFO_BRIEF_KERNEL 9

/* integral image, (h + 1) x (w + 1) int32, first row and column zero (cv::integral, CV_32S) */
static int32_t* sfo_integral(const uint8_t* img, int w, int h, int pitch) {
  int32_t* s = (int32_t*)calloc((size_t)(w + 1) * (h + 1), sizeof(int32_t));
  if (!s) return NULL;
  for (int y = 0; y < h; ++y) {
    int32_t run = 0;
    for (int x = 0; x < w; ++x) {
      run += img[(size_t)y * pitch + x];
      s[(size_t)(y + 1) * (w + 1) + x + 1] = s[(size_t)y * (w + 1) + x + 1] + run;
    }
  }
  return s;
}

/* brief.cpp smoothedSum(): the 9x9 box around (pt + offset), pt rounded half up.  A corner that rounds onto the
 * border limit (e.g. y = h - 28.3 -> h - 28) combined with an offset of +24 asks for row h + 1 of an integral image
 * that has rows 0..h: upstream that read is out of bounds; here it repeats the last row / column (h1, w1 = h + 1,
 * w + 1 are the integral image's sizes). */
static int32_t sfo_smoothed(const int32_t* s, int w1, int h1, int px, int py, int dx, int dy) {
  const int hk = SFO_BRIEF_KERNEL / 2;
  const int y = py + dy, x = px + dx;
  const int y1 = y + hk + 1 < h1 ? y + hk + 1 : h1 - 1, x1 = x + hk + 1 < w1 ? x + hk + 1 : w1 - 1;
  return s[(size_t)y1 * w1 + x1] - s[(size_t)y1 * w1 + x - hk] - s[(size_t)(y - hk) * w1 + x1] +
         s[(size_t)(y - hk) * w1 + x - hk];
}

int sfo_extract_keyframe(const uint8_t* image, int32_t width, int32_t height, int32_t pitch, const sf_keypoint* kpts,
                         const float* right_x, const uint8_t* status, int32_t n, const sf_stereo_camera* cam,
                         const int8_t* tests, int32_t bytes, uint8_t* desc_out, float* xyz_out, sf_keypoint* kp_out,
                         int32_t* rows_out) {
  if (!image || !cam || !tests || !rows_out || n < 0 || bytes < 1 || bytes > 64 || width < 1 || height < 1 ||
      pitch < width)
    return SF_EINVAL;
  for (int t = 0; t < 8 * bytes * 4; ++t)
    if (tests[t] < -SFO_BRIEF_PATCH / 2 || tests[t] > SFO_BRIEF_PATCH / 2) return SF_ERANGE;
  int32_t* s = sfo_integral(image, width, height, pitch);
  if (!s) return SF_ENOMEM;
  const int w1 = width + 1;
  const int border = SFO_BRIEF_PATCH / 2 + SFO_BRIEF_KERNEL / 2;
  const int filter = cam->min_depth > 0.0f || cam->max_depth > 0.0f;   /* myRegistrationVis.cpp:384 */
  const float nanf_ = nanf("");
  int out = 0;
  for (int i = 0; i < n; ++i) {
    const sf_keypoint k = kpts[i];
    /* KeyPointsFilter::runByImageBorder: Rect(border, border, w - 2 border, h - 2 border).contains(pt) */
    if (!(k.x >= (float)border && k.x < (float)(width - border) && k.y >= (float)border && k.y < (float)(height - border)))
      continue;
    const int px = (int)(k.x + 0.5f), py = (int)(k.y + 0.5f);
    uint8_t d[64];
    for (int b = 0; b < bytes; ++b) {
      unsigned v = 0;
      for (int t = 0; t < 8; ++t) {
        const int8_t* q = tests + (size_t)(8 * b + t) * 4;   /* x1, y1, x2, y2 */
        v = (v << 1) | (unsigned)(sfo_smoothed(s, w1, height + 1, px, py, q[0], q[1]) < sfo_smoothed(s, w1, height + 1, px, py, q[2], q[3]));
      }
      d[b] = (uint8_t)v;
    }
    /* util3d::generateKeypoints3DStereo */
    float p[3] = {nanf_, nanf_, nanf_};
    if (right_x && (!status || status[i])) {
      const float disparity = k.x - right_x[i];
      if (disparity != 0.0f && disparity > 0.0f && cam->baseline > 0.0f && cam->fx > 0.0f) {   /* projectDisparityTo3D */
        float c = 0.0f;
        if (cam->cx_right > 0.0f && cam->cx > 0.0f) c = cam->cx_right - cam->cx;
        const float W = cam->baseline / (disparity + c);
        const float x = (k.x - cam->cx) * W, y = (k.y - cam->cy) * W, z = cam->fx * W;
        if (isfinite(x) && isfinite(y) && isfinite(z) && (cam->min_depth < 0.0f || z > cam->min_depth) &&
            (cam->max_depth <= 0.0f || z <= cam->max_depth)) {
          const float* L = cam->local_transform;   /* util3d::transformPoint, skipped for the identity */
          int ident = 1;
          for (int e = 0; e < 12; ++e) ident = ident && (L[e] == ((e == 0 || e == 5 || e == 10) ? 1.0f : 0.0f));
          if (ident) {
            p[0] = x; p[1] = y; p[2] = z;
          } else {
            p[0] = ((L[0] * x + L[1] * y) + L[2] * z) + L[3];
            p[1] = ((L[4] * x + L[5] * y) + L[6] * z) + L[7];
            p[2] = ((L[8] * x + L[9] * y) + L[10] * z) + L[11];
          }
        }
      }
    }
    if (filter && !(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;   /* :397-409 */
    if (desc_out) memcpy(desc_out + (size_t)out * bytes, d, (size_t)bytes);
    if (xyz_out) memcpy(xyz_out + (size_t)out * 3, p, sizeof p);
    if (kp_out) kp_out[out] = k;
    ++out;
  }
  free(s);
  *rows_out = out;
  return SF_OK;
}
