// k_lk.hip -- stereo correspondence of the corners: the right-image position of every
// corner, what rtabmap's Feature2D::generateKeypoints3D obtains from StereoOpticalFlow::computeCorrespondences
// (cv::calcOpticalFlowPyrLK + disparity gate) for the reference's getFeaturesImpl (myRegistrationVis.cpp:382, called
// from stereoCamGeometricTools.cpp:100-120).  Arithmetic and order are those of the CPU restatement the tests compare
// against (deviation 17 names the one place where OpenCV's result depends on its build: float
// versus integer accumulation of A and b); compiled with -ffp-contract=off.
//
//   k_lk_pyr_down   one thread per output pixel of a pyramid level, both images in one launch (blockIdx.z): 5 x 5
//                   Gaussian in integers, BORDER_REFLECT_101.  Levels are small (361 KB at level 0 of a 752 x 480
//                   image, a quarter of that per level down): the launches are latency, not bandwidth.
//   k_lk_track      one 64-lane workgroup per corner, all pyramid levels inside the kernel (the corners are
//                   independent, the levels of one corner are not).  Per level: the (w + 3) x (h + 3) patch of the
//                   first image goes to LDS once; its Scharr derivatives are computed THERE for the (w + 1) x (h + 1)
//                   positions the bilinear taps touch (no derivative planes in HBM: 16 B per pixel and level saved,
//                   and one launch per level); lanes own window elements; A and b are exact integer sums reduced
//                   across the wavefront (DPP row scans + v_readlane for windows of <= 64 pixels), so the result
//                   does not depend on the reduction order.  The second image around the level's start position
//                   (+-8 x +-3 pixels beyond the window) is fetched into LDS together with the first image's patch:
//                   ONE trip to memory per level, and the chain of dependent Newton steps (<= 30 per level) runs
//                   from LDS; a step that leaves the region re-centres it (same pixels, so the same result).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "sf_internal.hpp"

namespace {

constexpr int LK_MAX_LEVELS = 16;
constexpr int LK_W_BITS = 14;
constexpr int LK_REGION_X = 8, LK_REGION_Y = 3;

struct LkLevel {
  const uint8_t* l;
  const uint8_t* r;
  int w, h, pitch, pad_;
};
struct LkLevels {
  LkLevel v[LK_MAX_LEVELS];
  int n;
  int per_image;                  // batch (blockIdx.y = image): corners / outputs of consecutive images this many apart
  size_t stride0, stride_pyr;     // bytes between consecutive images at level 0 / between their pyramid blocks
  const int32_t* d_n;             // batch: corners of every image (device); null: the scalar n
};

__device__ __forceinline__ int border_101(int p, int len) {       // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    p = p < 0 ? -p : 2 * len - 2 - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

__global__ __launch_bounds__(256) void k_lk_pyr_down(const uint8_t* __restrict__ src_l, const uint8_t* __restrict__ src_r,
                                                     int w, int h, int pitch, uint8_t* __restrict__ dst_l,
                                                     uint8_t* __restrict__ dst_r, int dw, int dh, size_t src_stride,
                                                     size_t dst_stride) {
  // blockIdx.z = 2 * image + (0: left, 1: right)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const uint8_t* src = ((blockIdx.z & 1) ? src_r : src_l) + (blockIdx.z >> 1) * src_stride;
  uint8_t* dst = ((blockIdx.z & 1) ? dst_r : dst_l) + (blockIdx.z >> 1) * dst_stride;
  int cx[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) cx[i] = border_101(2 * x + i - 2, w);
  int sum = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)border_101(2 * y + j - 2, h) * pitch;
    const int r = (int)row[cx[0]] + (int)row[cx[4]] + 4 * ((int)row[cx[1]] + (int)row[cx[3]]) + 6 * (int)row[cx[2]];
    sum += (j == 0 || j == 4) ? r : (j == 2 ? 6 * r : 4 * r);
  }
  dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// 32-bit sum over the wavefront without the LDS crossbar: a shifted-add scan inside each row of 16 lanes on the DPP
// path (lanes without a source add 0), then the four row totals through v_readlane -- the result is wave uniform.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  return (__builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31)) +
         (__builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63));
}
// Up to 64 window elements (one per lane) the sums of products stay below 2^31 -- |I_x|, |I_y| <= 4080 (Scharr of 8-bit
// data), |J - I| <= 8160 (5 fractional bits): 64 * 8160 * 4080 = 2 130 739 200 -- and take this path.
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct LkWeights { int w00, w01, w10, w11; };
__device__ __forceinline__ LkWeights lk_weights(float a, float b) {
  LkWeights w;
  w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << LK_W_BITS));
  w.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << LK_W_BITS));
  w.w10 = __float2int_rn((1.f - a) * b * (float)(1 << LK_W_BITS));
  w.w11 = (1 << LK_W_BITS) - w.w00 - w.w01 - w.w10;
  return w;
}

// SMALL: windows of <= 64 pixels (rtabmap's 15 x 3): one element per lane, kept in registers, 32-bit sums.
template <bool SMALL>
__global__ __launch_bounds__(64) void k_lk_track(const LkLevels P, const sf_keypoint* __restrict__ kp, int n, int ww, int wh,
                                                 int max_count, double eps2, float min_eig_thr, float min_disp,
                                                 float max_disp, float* __restrict__ xy_out, uint8_t* __restrict__ st_out,
                                                 float* __restrict__ rx_out, float* __restrict__ err_out) {
  extern __shared__ unsigned char lk_smem[];
  const int p = blockIdx.x;
  const int img = blockIdx.y;
  if (P.d_n) n = min(n, P.d_n[img]);
  if (p >= n) return;
  {
    const size_t o = (size_t)img * P.per_image;
    kp += o; xy_out += 2 * o; st_out += o;
    if (rx_out) rx_out += o;
    if (err_out) err_out += o;
  }
  const int lane = threadIdx.x;
  const int pw = ww + 3, ph = wh + 3;          // patch of the first image: window + bilinear tap + derivative ring
  const int dw = ww + 1, dh = wh + 1;          // positions the bilinear taps touch
  const int area = ww * wh;
  short2* der = reinterpret_cast<short2*>(lk_smem);
  short2* dIw = der + dw * dh;
  short* Iw = reinterpret_cast<short*>(dIw + area);
  unsigned char* patch = reinterpret_cast<unsigned char*>(Iw + ((area + 1) & ~1));
  // the second image around the level's start position: the Newton steps read it from LDS and only a step that
  // leaves it (more than LK_REGION_X / LK_REGION_Y pixels from the start) costs another trip to memory
  const int rw = dw + 2 * LK_REGION_X, rh = dh + 2 * LK_REGION_Y;
  unsigned char* region = patch + ((pw * ph + 3) & ~3);
  const int ey = SMALL ? lane / ww : 0, ex = SMALL ? lane - ey * ww : 0;      // this lane's window element (SMALL)
  const bool act = lane < area;
  const int roff = ey * rw + ex;
  int rI = 0, rdx = 0, rdy = 0;

  const float kx = kp[p].x, ky = kp[p].y;
  const float half_x = (float)(ww - 1) * 0.5f, half_y = (float)(wh - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const int max_level = P.n - 1;
  int st = 1;
  float er = 0.0f, nx = 0.0f, ny = 0.0f;
  for (int level = max_level; level >= 0; --level) {
    LkLevel L = P.v[level];
    {
      const size_t o = (size_t)img * (level == 0 ? P.stride0 : P.stride_pyr);
      L.l += o; L.r += o;
    }
    const float sc = (float)(1. / (double)(1 << level));
    float px = kx * sc, py = ky * sc;
    float qx, qy;
    if (level == max_level) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
    nx = qx; ny = qy;
    px -= half_x; py -= half_y;
    // (not finite or beyond +-2^30 = outside: x86's float -> int conversion in OpenCV gives INT_MIN there)
    const bool p_bad = !(fabsf(px) < 1073741824.f && fabsf(py) < 1073741824.f);
    const int ipx = p_bad ? 0 : (int)floorf(px), ipy = p_bad ? 0 : (int)floorf(py);
    if (p_bad || ipx < -ww || ipx >= L.w || ipy < -wh || ipy >= L.h) {
      if (level == 0) { st = 0; er = 0.0f; }
      continue;
    }
    __syncthreads();                                       // (the previous level's readers of the patches are done)
    for (int e = lane; e < pw * ph; e += 64) {
      const int yy = e / pw, xx = e - yy * pw;
      patch[e] = L.l[(size_t)border_101(ipy - 1 + yy, L.h) * L.pitch + border_101(ipx - 1 + xx, L.w)];
    }
    int rx0 = (int)floorf(qx - half_x) - LK_REGION_X, ry0 = (int)floorf(qy - half_y) - LK_REGION_Y;
    for (int e = lane; e < rw * rh; e += 64) {             // (in flight together with the patch above)
      const int yy = e / rw, xx = e - yy * rw;
      region[e] = L.r[(size_t)border_101(ry0 + yy, L.h) * L.pitch + border_101(rx0 + xx, L.w)];
    }
    __syncthreads();
    // calcSharrDeriv at the positions inside the image (REFLECT_101 taps are in the patch already), 0 outside
    for (int e = lane; e < dw * dh; e += 64) {
      const int yy = e / dw, xx = e - yy * dw;
      const unsigned char* r0 = patch + yy * pw + xx;      // rows yy, yy + 1, yy + 2 of the patch = y - 1, y, y + 1
      const unsigned char* r1 = r0 + pw;
      const unsigned char* r2 = r1 + pw;
      const int s0 = ((int)r0[0] + (int)r2[0]) * 3 + (int)r1[0] * 10, s2 = ((int)r0[2] + (int)r2[2]) * 3 + (int)r1[2] * 10;
      const int d0 = (int)r2[0] - (int)r0[0], d1 = (int)r2[1] - (int)r0[1], d2 = (int)r2[2] - (int)r0[2];
      const bool inside = (unsigned)(ipx + xx) < (unsigned)L.w && (unsigned)(ipy + yy) < (unsigned)L.h;
      short2 v;
      v.x = inside ? (short)(s2 - s0) : (short)0;
      v.y = inside ? (short)((d2 + d0) * 3 + d1 * 10) : (short)0;
      der[e] = v;
    }
    __syncthreads();
    LkWeights W = lk_weights(px - (float)ipx, py - (float)ipy);
    long long a11 = 0, a12 = 0, a22 = 0;
    if constexpr (SMALL) {
      int s11 = 0, s12 = 0, s22 = 0;
      if (act) {
        const unsigned char* q0 = patch + (ey + 1) * pw + ex + 1;
        const unsigned char* q1 = q0 + pw;
        rI = descale((int)q0[0] * W.w00 + (int)q0[1] * W.w01 + (int)q1[0] * W.w10 + (int)q1[1] * W.w11, LK_W_BITS - 5);
        const short2 e00 = der[ey * dw + ex], e01 = der[ey * dw + ex + 1], e10 = der[(ey + 1) * dw + ex], e11 = der[(ey + 1) * dw + ex + 1];
        rdx = descale((int)e00.x * W.w00 + (int)e01.x * W.w01 + (int)e10.x * W.w10 + (int)e11.x * W.w11, LK_W_BITS);
        rdy = descale((int)e00.y * W.w00 + (int)e01.y * W.w01 + (int)e10.y * W.w10 + (int)e11.y * W.w11, LK_W_BITS);
        s11 = rdx * rdx; s12 = rdx * rdy; s22 = rdy * rdy;
      }
      a11 = wave_sum_i32(s11); a12 = wave_sum_i32(s12); a22 = wave_sum_i32(s22);
    } else {
      for (int e = lane; e < area; e += 64) {
        const int y = e / ww, x = e - y * ww;
        const unsigned char* q0 = patch + (y + 1) * pw + x + 1;
        const unsigned char* q1 = q0 + pw;
        const int ival = descale((int)q0[0] * W.w00 + (int)q0[1] * W.w01 + (int)q1[0] * W.w10 + (int)q1[1] * W.w11, LK_W_BITS - 5);
        const short2 e00 = der[y * dw + x], e01 = der[y * dw + x + 1], e10 = der[(y + 1) * dw + x], e11 = der[(y + 1) * dw + x + 1];
        const int ixval = descale((int)e00.x * W.w00 + (int)e01.x * W.w01 + (int)e10.x * W.w10 + (int)e11.x * W.w11, LK_W_BITS);
        const int iyval = descale((int)e00.y * W.w00 + (int)e01.y * W.w01 + (int)e10.y * W.w10 + (int)e11.y * W.w11, LK_W_BITS);
        Iw[e] = (short)ival;
        dIw[e] = make_short2((short)ixval, (short)iyval);
        a11 += (long long)(ixval * ixval); a12 += (long long)(ixval * iyval); a22 += (long long)(iyval * iyval);
      }
      a11 = wave_sum(a11); a12 = wave_sum(a12); a22 = wave_sum(a22);
    }
    const float A11 = (float)a11 * FLT_SCALE, A12 = (float)a12 * FLT_SCALE, A22 = (float)a22 * FLT_SCALE;
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
      const bool q_bad = !(fabsf(qx) < 1073741824.f && fabsf(qy) < 1073741824.f);
      const int iqx = q_bad ? 0 : (int)floorf(qx), iqy = q_bad ? 0 : (int)floorf(qy);
      if (q_bad || iqx < -ww || iqx >= L.w || iqy < -wh || iqy >= L.h) {
        if (level == 0) st = 0;
        break;
      }
      if (iqx < rx0 || iqx + dw > rx0 + rw || iqy < ry0 || iqy + dh > ry0 + rh) {     // left the region: centre it here
        rx0 = iqx - LK_REGION_X; ry0 = iqy - LK_REGION_Y;
        __syncthreads();
        for (int e = lane; e < rw * rh; e += 64) {
          const int yy = e / rw, xx = e - yy * rw;
          region[e] = L.r[(size_t)border_101(ry0 + yy, L.h) * L.pitch + border_101(rx0 + xx, L.w)];
        }
        __syncthreads();
      }
      W = lk_weights(qx - (float)iqx, qy - (float)iqy);
      long long b1s = 0, b2s = 0;
      const unsigned char* rbase = region + (iqy - ry0) * rw + (iqx - rx0);
      if constexpr (SMALL) {
        int t1 = 0, t2 = 0;
        if (act) {
          const unsigned char* q0 = rbase + roff;
          const unsigned char* q1 = q0 + rw;
          const int diff = descale((int)q0[0] * W.w00 + (int)q0[1] * W.w01 + (int)q1[0] * W.w10 + (int)q1[1] * W.w11, LK_W_BITS - 5) - rI;
          t1 = diff * rdx; t2 = diff * rdy;
        }
        b1s = wave_sum_i32(t1); b2s = wave_sum_i32(t2);
      } else {
        for (int e = lane; e < area; e += 64) {
          const int y = e / ww, x = e - y * ww;
          const unsigned char* q0 = rbase + y * rw + x;
          const unsigned char* q1 = q0 + rw;
          const int diff = descale((int)q0[0] * W.w00 + (int)q0[1] * W.w01 + (int)q1[0] * W.w10 + (int)q1[1] * W.w11, LK_W_BITS - 5) - (int)Iw[e];
          const short2 d = dIw[e];
          b1s += (long long)(diff * (int)d.x); b2s += (long long)(diff * (int)d.y);
        }
        b1s = wave_sum(b1s); b2s = wave_sum(b2s);
      }
      const float b1 = (float)b1s * FLT_SCALE, b2 = (float)b2s * FLT_SCALE;
      const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
      qx += dx; qy += dy;
      nx = qx + half_x; ny = qy + half_y;
      if ((double)dx * (double)dx + (double)dy * (double)dy <= eps2) break;
      if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
        nx -= dx * 0.5f; ny -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
  }
  if (st) {                                               // StereoOpticalFlow::updateStatus [upstream rtabmap]
    const float disparity = kx - nx;
    if (disparity <= min_disp || disparity > max_disp) st = 0;
  }
  if (lane == 0) {
    xy_out[2 * p] = nx; xy_out[2 * p + 1] = ny;
    st_out[p] = (uint8_t)st;
    if (rx_out) rx_out[p] = nx;
    if (err_out) err_out[p] = er;
  }
}

}  // namespace

// n_img > 1: a batch -- image i at d_left / d_right + i * img_stride, its corners / outputs at + i * n entries, its corner
// count in d_n[i] (device; `n` is then the per-image capacity the grid is sized for).
int sf_launch_stereo_flow_batch(sf_context* c, const uint8_t* d_left, const uint8_t* d_right, size_t img_stride, int n_img,
                                int width, int height, int pitch, const sf_keypoint* d_kpts, int n, const int32_t* d_n,
                                const sf_stereo_flow_params* prm, float* d_right_xy, uint8_t* d_status, float* d_right_x,
                                float* d_err) {
  const int ww = prm->win_width, wh = prm->win_height;
  // buildOpticalFlowPyramid: the level whose successor would be <= winSize in either direction is the last
  int lw[LK_MAX_LEVELS], lh[LK_MAX_LEVELS];
  size_t off[LK_MAX_LEVELS];
  int nl = 0;
  size_t bytes = 0;
  {
    int w = width, h = height;
    for (int l = 0; l <= prm->max_level; ++l) {
      lw[l] = w; lh[l] = h;
      off[l] = bytes;
      if (l > 0) bytes += ((size_t)w * h + 15) & ~(size_t)15;
      nl = l + 1;
      w = (w + 1) / 2; h = (h + 1) / 2;
      if (w <= ww || h <= wh) break;
    }
  }
  int rc;
  if ((rc = sf_buf_reserve(c, c->lk_pyr, std::max<size_t>(2 * bytes * n_img, 16))) != SF_OK) return rc;
  uint8_t* base_l = (uint8_t*)c->lk_pyr.p;
  uint8_t* base_r = base_l + bytes * n_img;
  LkLevels P;
  P.n = nl;
  P.per_image = n;
  P.stride0 = img_stride;
  P.stride_pyr = bytes;
  P.d_n = d_n;
  for (int l = 0; l < nl; ++l) {
    P.v[l].l = l == 0 ? d_left : base_l + off[l];
    P.v[l].r = l == 0 ? d_right : base_r + off[l];
    P.v[l].w = lw[l]; P.v[l].h = lh[l]; P.v[l].pitch = l == 0 ? pitch : lw[l]; P.v[l].pad_ = 0;
  }
  for (int l = 1; l < nl; ++l) {
    const dim3 grid((lw[l] + 63) / 64, (lh[l] + 3) / 4, 2 * n_img);
    hipLaunchKernelGGL(k_lk_pyr_down, grid, dim3(256), 0, c->stream, P.v[l - 1].l, P.v[l - 1].r, lw[l - 1], lh[l - 1],
                       P.v[l - 1].pitch, (uint8_t*)P.v[l].l, (uint8_t*)P.v[l].r, lw[l], lh[l],
                       l == 1 ? img_stride : bytes, bytes);
  }
  const int max_count = std::min(std::max(prm->iterations, 0), 100);
  double eps = std::min(std::max(prm->epsilon, 0.0), 10.0);
  eps *= eps;
  const int area = ww * wh;
  const size_t smem = (size_t)(ww + 1) * (wh + 1) * 4 + (size_t)area * 4 + (size_t)((area + 1) & ~1) * 2 +
                      (((size_t)(ww + 3) * (wh + 3) + 3) & ~(size_t)3) +
                      (size_t)(ww + 1 + 2 * LK_REGION_X) * (wh + 1 + 2 * LK_REGION_Y);
  if (area <= 64)
    hipLaunchKernelGGL(k_lk_track<true>, dim3(n, n_img), dim3(64), smem, c->stream, P, d_kpts, n, ww, wh, max_count, eps,
                       prm->min_eig_threshold, prm->min_disparity, prm->max_disparity, d_right_xy, d_status, d_right_x, d_err);
  else
    hipLaunchKernelGGL(k_lk_track<false>, dim3(n, n_img), dim3(64), smem, c->stream, P, d_kpts, n, ww, wh, max_count, eps,
                       prm->min_eig_threshold, prm->min_disparity, prm->max_disparity, d_right_xy, d_status, d_right_x, d_err);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

int sf_launch_stereo_flow(sf_context* c, const uint8_t* d_left, const uint8_t* d_right, int width, int height, int pitch,
                          const sf_keypoint* d_kpts, int n, const sf_stereo_flow_params* prm, float* d_right_xy,
                          uint8_t* d_status, float* d_right_x, float* d_err) {
  return sf_launch_stereo_flow_batch(c, d_left, d_right, 0, 1, width, height, pitch, d_kpts, n, nullptr, prm, d_right_xy,
                                     d_status, d_right_x, d_err);
}
