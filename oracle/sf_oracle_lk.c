/* sf_oracle_lk.c -- CPU ORACLE (test infrastructure, never linked into the product) for the stereo correspondence of
 * the right-image position of every corner, as Feature2D::generateKeypoints3D [upstream
 * rtabmap] obtains it for the reference's getFeaturesImpl (myRegistrationVis.cpp:382, called from
 * stereoCamGeometricTools.cpp:100-120): StereoOpticalFlow::computeCorrespondences = cv::calcOpticalFlowPyrLK(left,
 * right, corners, winSize (Stereo/WinWidth, Stereo/WinHeight), Stereo/MaxLevel, {COUNT + EPS, Stereo/Iterations,
 * Stereo/Eps}, OPTFLOW_LK_GET_MIN_EIGENVALS, 1e-4), then status = 0 where the disparity left.x - right.x is
 * <= Stereo/MinDisparity or > Stereo/MaxDisparity.
 *
 * PARITY UNPINNED: cv::calcOpticalFlowPyrLK lives in OpenCV (video/lkpyramid.cpp, imgproc/pyramids.cpp), which is
 * neither in the reference tree nor installed here.  Restated from its published algorithm (OpenCV 3.x / 4.x, 8-bit
 * single-channel input, no initial flow):
 *   1. buildOpticalFlowPyramid: level 0 = the image, level l = pyrDown(level l - 1) (5 x 5 Gaussian (1 4 6 4 1)^2,
 *      (sum + 128) >> 8, size (w + 1) / 2 x (h + 1) / 2, BORDER_REFLECT_101); every level carries a REFLECT_101 border
 *      of winSize pixels; levels stop (maxLevel shrinks) once the NEXT size is <= winSize in either direction;
 *   2. calcSharrDeriv of every level of the FIRST image: int16 {Ix, Iy} = Scharr (3 10 3) x (-1 0 1) with
 *      REFLECT_101 at the image edges; outside the image the derivative is 0 (BORDER_CONSTANT);
 *   3. per corner, from the top level down (LKTrackerInvoker): window patch of the first image and its derivatives by
 *      bilinear interpolation in 14-bit fixed point (value keeps 5 fractional bits), structure matrix A, minimum
 *      eigenvalue test (level skipped below 1e-4, status cleared at level 0), then <= maxCount Newton steps on the
 *      second image: b = sum (J - I) dI, delta = -A^-1 b, stop on |delta|^2 <= eps^2, on leaving the image (status
 *      cleared at level 0) or on a two-step oscillation (< 0.01 px, position moved back by delta / 2);
 *   4. err = minimum eigenvalue of A / window area at level 0 (OPTFLOW_LK_GET_MIN_EIGENVALS).
 * Where OpenCV's result depends on its build -- A and b are accumulated in float (four SSE lanes) on x86 and in
 * 64-bit integers on NEON -- this file takes the EXACT integer sums (the NEON form) and converts once; the GPU kernel
 * (csrc/k_lk.hip) does the same and is compared byte for byte.  Compiled with -ffp-contract=off.                  */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sf_oracle.h"

static int border_101(int p, int len) {          /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

/* pyrDown of an 8-bit image: dst is ((w + 1) / 2) x ((h + 1) / 2), tightly packed */
void sfo_pyr_down(const uint8_t* src, int32_t w, int32_t h, int32_t pitch, uint8_t* dst) {
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  static const int k5[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      int sum = 0;
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = src + (size_t)border_101(2 * y + j - 2, h) * pitch;
        int r = 0;
        for (int i = 0; i < 5; ++i) r += k5[i] * row[border_101(2 * x + i - 2, w)];
        sum += k5[j] * r;
      }
      dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
    }
}

/* calcSharrDeriv: dst [h][w][2] int16 {Ix, Iy} */
void sfo_scharr_deriv(const uint8_t* src, int32_t w, int32_t h, int32_t pitch, int16_t* dst) {
  for (int y = 0; y < h; ++y) {
    const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * pitch;
    const uint8_t* r1 = src + (size_t)y * pitch;
    const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * pitch;
    for (int x = 0; x < w; ++x) {
      const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
#define SM(c) ((r0[c] + r2[c]) * 3 + r1[c] * 10)      /* vertical smoothing */
#define DF(c) (r2[c] - r0[c])                         /* vertical difference */
      dst[((size_t)y * w + x) * 2] = (int16_t)(SM(xr) - SM(xl));
      dst[((size_t)y * w + x) * 2 + 1] = (int16_t)((DF(xr) + DF(xl)) * 3 + DF(x) * 10);
#undef SM
#undef DF
    }
  }
}

typedef struct { uint8_t* img; int16_t* deriv; int w, h; } sfo_level;

static void level_free(sfo_level* L, int n, int keep0) {
  for (int l = 0; l < n; ++l) {
    if (l > 0 || !keep0) free(L[l].img);
    free(L[l].deriv);
  }
}

#define SFO_LK_MAX_LEVELS 16
#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* pixel of a pyramid level INCLUDING its REFLECT_101 border */
static inline int px_img(const sfo_level* L, int x, int y) { return L->img[(size_t)border_101(y, L->h) * L->w + border_101(x, L->w)]; }
static inline int px_der(const sfo_level* L, int x, int y, int c) {
  if ((unsigned)x >= (unsigned)L->w || (unsigned)y >= (unsigned)L->h) return 0;
  return L->deriv[((size_t)y * L->w + x) * 2 + c];
}

int sfo_stereo_correspondences(const uint8_t* left, const uint8_t* right, int32_t width, int32_t height, int32_t pitch,
                               const sf_keypoint* kpts, int32_t n, const sf_stereo_flow_params* prm, float* right_xy,
                               uint8_t* status, float* err, int32_t* levels_used) {
  if (!left || !right || width < 1 || height < 1 || pitch < width || n < 0 || !prm || (n > 0 && (!kpts || !right_xy || !status)))
    return SF_EINVAL;
  const int ww = prm->win_width, wh = prm->win_height;
  if (ww <= 2 || wh <= 2 || prm->max_level < 0 || prm->max_level >= SFO_LK_MAX_LEVELS) return SF_EINVAL;   /* (OpenCV asserts winSize > 2) */
  sfo_level LL[SFO_LK_MAX_LEVELS], RR[SFO_LK_MAX_LEVELS];
  memset(LL, 0, sizeof LL); memset(RR, 0, sizeof RR);
  /* level 0: packed copies (the pitch is dropped once) */
  int nl = 0;
  {
    int w = width, h = height;
    for (int l = 0; l <= prm->max_level; ++l) {
      LL[l].w = RR[l].w = w; LL[l].h = RR[l].h = h;
      LL[l].img = (uint8_t*)malloc((size_t)w * h); RR[l].img = (uint8_t*)malloc((size_t)w * h);
      LL[l].deriv = (int16_t*)malloc((size_t)w * h * 2 * sizeof(int16_t));
      nl = l + 1;
      if (!LL[l].img || !RR[l].img || !LL[l].deriv) { level_free(LL, nl, 0); level_free(RR, nl, 0); return SF_ENOMEM; }
      if (l == 0) {
        for (int y = 0; y < h; ++y) { memcpy(LL[0].img + (size_t)y * w, left + (size_t)y * pitch, (size_t)w); memcpy(RR[0].img + (size_t)y * w, right + (size_t)y * pitch, (size_t)w); }
      } else {
        sfo_pyr_down(LL[l - 1].img, LL[l - 1].w, LL[l - 1].h, LL[l - 1].w, LL[l].img);
        sfo_pyr_down(RR[l - 1].img, RR[l - 1].w, RR[l - 1].h, RR[l - 1].w, RR[l].img);
      }
      sfo_scharr_deriv(LL[l].img, w, h, w, LL[l].deriv);
      w = (w + 1) / 2; h = (h + 1) / 2;
      if (w <= ww || h <= wh) break;                   /* buildOpticalFlowPyramid returns this level as the last */
    }
  }
  const int max_level = nl - 1;
  if (levels_used) *levels_used = nl;
  int max_count = prm->iterations < 0 ? 0 : (prm->iterations > 100 ? 100 : prm->iterations);
  double eps = prm->epsilon < 0.0 ? 0.0 : (prm->epsilon > 10.0 ? 10.0 : prm->epsilon);
  eps *= eps;
  const float half_x = (float)(ww - 1) * 0.5f, half_y = (float)(wh - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float min_eig_thr = prm->min_eig_threshold;
  int16_t* Iw = (int16_t*)malloc((size_t)ww * wh * 3 * sizeof(int16_t));
  if (!Iw) { level_free(LL, nl, 0); level_free(RR, nl, 0); return SF_ENOMEM; }
  int16_t* dIw = Iw + (size_t)ww * wh;
  for (int p = 0; p < n; ++p) {
    uint8_t st = 1;
    float er = 0.0f;                                    /* (OpenCV leaves err unset when no level reaches the test) */
    float nx = 0.0f, ny = 0.0f;                         /* nextPts[p] */
    for (int level = max_level; level >= 0; --level) {
      const sfo_level* I = &LL[level]; const sfo_level* J = &RR[level];
      const float sc = (float)(1. / (1 << level));
      float px = kpts[p].x * sc, py = kpts[p].y * sc;
      float qx, qy;
      if (level == max_level) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
      nx = qx; ny = qy;
      px -= half_x; py -= half_y;
      /* a position that is not finite or beyond +-2^30 is OUTSIDE (what x86's float -> int conversion makes of it
         in OpenCV: INT_MIN, which fails the test below; spelled out because the conversion itself is undefined) */
      const int p_bad = !(fabsf(px) < 1073741824.f && fabsf(py) < 1073741824.f);
      const int ipx = p_bad ? 0 : (int)floorf(px), ipy = p_bad ? 0 : (int)floorf(py);
      if (p_bad || ipx < -ww || ipx >= I->w || ipy < -wh || ipy >= I->h) {
        if (level == 0) { st = 0; er = 0.0f; }
        continue;
      }
      float a = px - (float)ipx, b = py - (float)ipy;
      int iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
      int iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      int64_t iA11 = 0, iA12 = 0, iA22 = 0;
      for (int y = 0; y < wh; ++y)
        for (int x = 0; x < ww; ++x) {
          const int X = ipx + x, Y = ipy + y;
          const int ival = DESCALE(px_img(I, X, Y) * iw00 + px_img(I, X + 1, Y) * iw01 + px_img(I, X, Y + 1) * iw10 + px_img(I, X + 1, Y + 1) * iw11, W_BITS - 5);
          const int ixval = DESCALE(px_der(I, X, Y, 0) * iw00 + px_der(I, X + 1, Y, 0) * iw01 + px_der(I, X, Y + 1, 0) * iw10 + px_der(I, X + 1, Y + 1, 0) * iw11, W_BITS);
          const int iyval = DESCALE(px_der(I, X, Y, 1) * iw00 + px_der(I, X + 1, Y, 1) * iw01 + px_der(I, X, Y + 1, 1) * iw10 + px_der(I, X + 1, Y + 1, 1) * iw11, W_BITS);
          Iw[y * ww + x] = (int16_t)ival;
          dIw[(y * ww + x) * 2] = (int16_t)ixval;
          dIw[(y * ww + x) * 2 + 1] = (int16_t)iyval;
          iA11 += (int64_t)ixval * ixval; iA12 += (int64_t)ixval * iyval; iA22 += (int64_t)iyval * iyval;
        }
      const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
      er = min_eig;
      if (min_eig < min_eig_thr || D < 1.1920929e-07f) {
        if (level == 0) st = 0;
        continue;
      }
      D = 1.f / D;
      qx -= half_x; qy -= half_y;
      float pdx = 0.0f, pdy = 0.0f;
      for (int j = 0; j < max_count; ++j) {
        const int q_bad = !(fabsf(qx) < 1073741824.f && fabsf(qy) < 1073741824.f);
        const int iqx = q_bad ? 0 : (int)floorf(qx), iqy = q_bad ? 0 : (int)floorf(qy);
        if (q_bad || iqx < -ww || iqx >= J->w || iqy < -wh || iqy >= J->h) {
          if (level == 0) st = 0;
          break;
        }
        a = qx - (float)iqx; b = qy - (float)iqy;
        iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
        iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t ib1 = 0, ib2 = 0;
        for (int y = 0; y < wh; ++y)
          for (int x = 0; x < ww; ++x) {
            const int X = iqx + x, Y = iqy + y;
            const int diff = DESCALE(px_img(J, X, Y) * iw00 + px_img(J, X + 1, Y) * iw01 + px_img(J, X, Y + 1) * iw10 + px_img(J, X + 1, Y + 1) * iw11, W_BITS - 5) - Iw[y * ww + x];
            ib1 += (int64_t)diff * dIw[(y * ww + x) * 2];
            ib2 += (int64_t)diff * dIw[(y * ww + x) * 2 + 1];
          }
        const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        qx += dx; qy += dy;
        nx = qx + half_x; ny = qy + half_y;
        if ((double)dx * (double)dx + (double)dy * (double)dy <= eps) break;
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
          nx -= dx * 0.5f; ny -= dy * 0.5f;
          break;
        }
        pdx = dx; pdy = dy;
      }
    }
    /* StereoOpticalFlow::updateStatus [upstream rtabmap Stereo.cpp] */
    if (st) {
      const float disparity = kpts[p].x - nx;
      if (disparity <= prm->min_disparity || disparity > prm->max_disparity) st = 0;
    }
    right_xy[2 * p] = nx; right_xy[2 * p + 1] = ny;
    status[p] = st;
    if (err) err[p] = er;
  }
  free(Iw);
  level_free(LL, nl, 0); level_free(RR, nl, 0);
  return SF_OK;
}
