/* sf_oracle_gftt.c -- CPU ORACLE (test infrastructure, never linked into the product) for the corner detector of
 * what Feature2D::generateKeypoints does for the reference's default feature type
 * (rtabmap GFTT/BRIEF, Vis/FeatureType 6 -> cv::GFTTDetector -> cv::goodFeaturesToTrack; called from
 * myRegistrationVis.cpp:281-283).
 *
 * PARITY UNPINNED: cv::goodFeaturesToTrack lives in OpenCV (imgproc/featureselect.cpp, corner.cpp), not in the
 * reference tree and not installed here.  Restated from its published algorithm (OpenCV 3.x, no mask, no Harris):
 *   1. cornerMinEigenVal(blockSize = 3, ksize = 3): Sobel derivatives scaled by 1 / (4 * 3 * 255), their products
 *      summed over the 3 x 3 block (unnormalised box filter), eig = (a + c) - sqrt((a - c)^2 + b^2) with
 *      a = sum(dx dx) / 2, b = sum(dx dy), c = sum(dy dy) / 2; borders BORDER_REFLECT_101 for both filters;
 *   2. everything not above qualityLevel * max(eig) is zeroed; corners = pixels of rows 1..h-2, columns 1..w-2 that
 *      are non-zero and equal to the maximum of their 3 x 3 neighbourhood (cv::dilate);
 *   3. sorted by decreasing response, ties by decreasing address (featureselect.cpp greaterThanPtr);
 *   4. taken in that order while no already taken corner is closer than minDistance (grid of cvRound(minDistance)
 *      cells), until maxCorners are taken;
 *   5. keypoints: pt = (x, y), size = blockSize, angle = -1, response = 0, octave = 0, class_id = -1.
 * Where OpenCV's float summation order depends on its build (the box filter keeps RUNNING column sums; the separable
 * Sobel has SIMD paths), this file fixes one order, spelled out below; the GPU kernels (csrc/k_gftt.hip) use the same
 * one and are compared byte for byte.  Compiled with -ffp-contract=off.                                            */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sf_oracle.h"

static int refl(int i, int n) {          /* BORDER_REFLECT_101, n >= 2 */
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

typedef struct { float v; int32_t idx; } sfo_cand;
static int sfo_cand_cmp(const void* pa, const void* pb) {
  const sfo_cand* a = (const sfo_cand*)pa; const sfo_cand* b = (const sfo_cand*)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}

int sfo_detect_corners(const uint8_t* image, int32_t width, int32_t height, int32_t pitch, int32_t max_corners,
                       double quality_level, double min_distance, sf_keypoint* kpts_out, int32_t cap, int32_t* n_out,
                       float* eig_out) {
  if (!image || !n_out || width < 3 || height < 3 || pitch < width || quality_level <= 0.0 || min_distance < 0.0 || cap < 0)
    return SF_EINVAL;
  const int w = width, h = height;
  const size_t np = (size_t)w * h;
  float* dxx = (float*)malloc(np * sizeof(float));
  float* dxy = (float*)malloc(np * sizeof(float));
  float* dyy = (float*)malloc(np * sizeof(float));
  float* eig = (float*)malloc(np * sizeof(float));
  sfo_cand* cand = (sfo_cand*)malloc(np * sizeof(sfo_cand));
  if (!dxx || !dxy || !dyy || !eig || !cand) { free(dxx); free(dxy); free(dyy); free(eig); free(cand); return SF_ENOMEM; }
  const double scale = 1.0 / ((double)(1 << 2) * 3.0 * 255.0);
  const float s1 = (float)(1.0 * scale), s2 = (float)(2.0 * scale);
#define PX(yy, xx) ((float)image[(size_t)refl((yy), h) * pitch + refl((xx), w)])
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      /* Dx: row derivative r(y) = p(x + 1) - p(x - 1), smoothed down the column with (s, 2s, s) */
      const float r0 = PX(y, x + 1) - PX(y, x - 1), ru = PX(y - 1, x + 1) - PX(y - 1, x - 1),
                  rd = PX(y + 1, x + 1) - PX(y + 1, x - 1);
      const float dx = s2 * r0 + s1 * (ru + rd);
      /* Dy: row smoothing c(y) = 2s p(x) + s (p(x - 1) + p(x + 1)), then c(y + 1) - c(y - 1) */
      const float cu = s2 * PX(y - 1, x) + s1 * (PX(y - 1, x - 1) + PX(y - 1, x + 1));
      const float cd = s2 * PX(y + 1, x) + s1 * (PX(y + 1, x - 1) + PX(y + 1, x + 1));
      const float dy = cd - cu;
      dxx[(size_t)y * w + x] = dx * dx; dxy[(size_t)y * w + x] = dx * dy; dyy[(size_t)y * w + x] = dy * dy;
    }
#undef PX
  float vmax = 0.0f;
#define BOX(P, yy, xx) ((((P)[(size_t)refl((yy) - 1, h) * w + refl((xx) - 1, w)] + (P)[(size_t)refl((yy) - 1, h) * w + (xx)]) + (P)[(size_t)refl((yy) - 1, h) * w + refl((xx) + 1, w)]) + \
                        (((P)[(size_t)(yy) * w + refl((xx) - 1, w)] + (P)[(size_t)(yy) * w + (xx)]) + (P)[(size_t)(yy) * w + refl((xx) + 1, w)])) + \
                       (((P)[(size_t)refl((yy) + 1, h) * w + refl((xx) - 1, w)] + (P)[(size_t)refl((yy) + 1, h) * w + (xx)]) + (P)[(size_t)refl((yy) + 1, h) * w + refl((xx) + 1, w)])
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float a = (BOX(dxx, y, x)) * 0.5f, b = BOX(dxy, y, x), c = (BOX(dyy, y, x)) * 0.5f;
      const float e = (a + c) - sqrtf((a - c) * (a - c) + b * b);
      eig[(size_t)y * w + x] = e;
      if (e > vmax) vmax = e;
    }
#undef BOX
  if (eig_out) memcpy(eig_out, eig, np * sizeof(float));
  const float thr = (float)((double)vmax * quality_level);
  int nc = 0;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const float v = eig[(size_t)y * w + x];
      if (!(v > thr)) continue;                       /* THRESH_TOZERO: zero, hence not a corner */
      float m = 0.0f;                                 /* (thresholded neighbours are 0; v > thr >= 0) */
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const float u = eig[(size_t)(y + dy) * w + x + dx];
          const float t = u > thr ? u : 0.0f;
          if (t > m) m = t;
        }
      if (v == m) { cand[nc].v = v; cand[nc].idx = y * w + x; ++nc; }
    }
  qsort(cand, (size_t)nc, sizeof(sfo_cand), sfo_cand_cmp);
  int out = 0;
  if (min_distance >= 1.0) {
    const int cell = (int)lrint(min_distance);        /* cvRound */
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    int32_t* head = (int32_t*)malloc((size_t)gw * gh * sizeof(int32_t));
    int32_t* next = (int32_t*)malloc((size_t)(nc > 0 ? nc : 1) * sizeof(int32_t));
    int32_t* px = (int32_t*)malloc((size_t)(nc > 0 ? nc : 1) * 2 * sizeof(int32_t));
    if (!head || !next || !px) { free(head); free(next); free(px); free(dxx); free(dxy); free(dyy); free(eig); free(cand); return SF_ENOMEM; }
    for (int i = 0; i < gw * gh; ++i) head[i] = -1;
    const float md2 = (float)(min_distance * min_distance);
    for (int i = 0; i < nc; ++i) {
      if (max_corners > 0 && out >= max_corners) break;
      const int y = cand[i].idx / w, x = cand[i].idx - y * w;
      const int cx = x / cell, cy = y / cell;
      int x1 = cx - 1, y1 = cy - 1, x2 = cx + 1, y2 = cy + 1;
      if (x1 < 0) x1 = 0;
      if (y1 < 0) y1 = 0;
      if (x2 > gw - 1) x2 = gw - 1;
      if (y2 > gh - 1) y2 = gh - 1;
      int good = 1;
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx = x1; xx <= x2 && good; ++xx)
          for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
            const float ddx = (float)(x - px[2 * j]), ddy = (float)(y - px[2 * j + 1]);
            if (ddx * ddx + ddy * ddy < md2) { good = 0; break; }
          }
      if (!good) continue;
      px[2 * out] = x; px[2 * out + 1] = y;
      next[out] = head[cy * gw + cx]; head[cy * gw + cx] = out;
      if (out < cap && kpts_out) {
        sf_keypoint k; k.x = (float)x; k.y = (float)y; k.size = 3.0f; k.angle = -1.0f; k.response = 0.0f; k.octave = 0; k.class_id = -1;
        kpts_out[out] = k;
      }
      ++out;
    }
    free(head); free(next); free(px);
  } else {
    for (int i = 0; i < nc; ++i) {
      if (max_corners > 0 && out >= max_corners) break;
      const int y = cand[i].idx / w, x = cand[i].idx - y * w;
      if (out < cap && kpts_out) {
        sf_keypoint k; k.x = (float)x; k.y = (float)y; k.size = 3.0f; k.angle = -1.0f; k.response = 0.0f; k.octave = 0; k.class_id = -1;
        kpts_out[out] = k;
      }
      ++out;
    }
  }
  free(dxx); free(dxy); free(dyy); free(eig); free(cand);
  *n_out = out;
  return SF_OK;
}
