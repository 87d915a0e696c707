// k_extract.hip -- SURVEY.md section 8 row f3: the features of one stereo keyframe for given corners, written
// straight into the device-resident keyframe store.
//
// What it replaces: RegistrationVis::getFeaturesImpl (myRegistrationVis.cpp:343-436) as called by
// StereoCamGeometricTools::getFeaturesAndDescriptor (stereoCamGeometricTools.cpp:100-120):
//   :343-354  descriptors for the given keypoints      -> BRIEF from the integral image of the left image
//   :356-383  3D keypoints of the stereo pair          -> disparity projection + local transform
//   :384-425  drop keypoints without a finite 3D point when a depth range is set
// The bodies of generateDescriptors / generateKeypoints3D are rtabmap / opencv_contrib code that is not in the
// reference tree; the algorithm restated here is documented next to the CPU restatement the tests compare with.
// Arithmetic: the float operations in the order written, no contraction (this file is compiled with
// -ffp-contract=off), IEEE division -- the outputs are compared byte for byte.
//
// Kernels (one keyframe = a few hundred to a few thousand corners of one 752 x 480 .. 1280 x 720 image):
//   k_integral_rows   one workgroup per image row: wavefront scans of 256-pixel segments, running carry
//   k_integral_cols   one thread per column: the running column sum (coalesced across the row)
//   k_extract_points  one thread per (corner, descriptor byte): 8 tests = 64 integral-image reads; byte 0's thread
//                     also does the border test and the 3D point
//   k_extract_commit  ONE workgroup: order-preserving compaction of the kept corners (block scan of the keep
//                     flags) into the store slot -- descriptor words, xyz, reduced keypoints, meta -- and into the
//                     optional wire copies
// The image is HBM-resident input (W*H bytes read once by the row pass, 4 (W+1)(H+1) bytes of integral image written
// and re-read through L2 by the tests); at these sizes every kernel is launch-latency-bound, which is why the
// per-keyframe work is three short launches and not a pipeline.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sf_internal.hpp"

namespace {

constexpr int BRIEF_PATCH = 48;
constexpr int BRIEF_KERNEL = 9;
constexpr int BRIEF_BORDER = BRIEF_PATCH / 2 + BRIEF_KERNEL / 2;   // KeyPointsFilter::runByImageBorder margin

// A batch of keyframes in one launch sequence (blockIdx.y = image, except k_extract_commit: blockIdx.x): images
// img_stride bytes apart, integral images s_stride entries apart, per-corner arrays per_image entries apart, the corner
// count of every image in d_n (device; null = the scalar n of the single-image call).
struct ExtractBatch {
  size_t img_stride, s_stride;
  int per_image;
  const int32_t* d_n;
};

// S is (h + 1) x (w + 1); this pass leaves ROW prefix sums in rows 1..h and zeroes row 0 / column 0.
__global__ void __launch_bounds__(256)
k_integral_rows(const uint8_t* __restrict__ img, int w, int h, int pitch, int32_t* __restrict__ S, ExtractBatch B) {
  __shared__ int wave_sum[4];
  img += blockIdx.y * B.img_stride;
  S += blockIdx.y * B.s_stride;
  const int y = blockIdx.x;          // 0..h: row y of S
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* row = S + (size_t)y * (w + 1);
  if (y == 0) {
    for (int x = tid; x <= w; x += 256) row[x] = 0;
    return;
  }
  const uint8_t* src = img + (size_t)(y - 1) * pitch;
  if (tid == 0) row[0] = 0;
  int carry = 0;
  for (int base = 0; base < w; base += 256) {
    const int x = base + tid;
    int v = x < w ? (int)src[x] : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(v, off);
      if (lane >= off) v += u;
    }
    if (lane == 63) wave_sum[wave] = v;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = wave_sum[q];
      if (q < wave) before += s;
      total += s;
    }
    if (x < w) row[x + 1] = carry + before + v;
    carry += total;
    __syncthreads();
  }
}

// One thread per column; the rows are taken 16 at a time so that 16 loads are in flight per dependent step (a
// load-add-store per row was ~480 serial L2 round trips: 120 of the call's 154 us at 752 x 480).
__global__ void __launch_bounds__(64)
k_integral_cols(int w, int h, int32_t* __restrict__ S, ExtractBatch B) {
  const int x = blockIdx.x * 64 + threadIdx.x;   // column 1..w of S
  if (x < 1 || x > w) return;
  S += blockIdx.y * B.s_stride;
  const size_t pitch = (size_t)(w + 1);
  int32_t run = 0;
  int y = 1;
  for (; y + 15 <= h; y += 16) {
    int32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = S[(size_t)(y + u) * pitch + x];
#pragma unroll
    for (int u = 0; u < 16; ++u) { run += v[u]; v[u] = run; }
#pragma unroll
    for (int u = 0; u < 16; ++u) S[(size_t)(y + u) * pitch + x] = v[u];
  }
  for (; y <= h; ++y) {
    int32_t* p = S + (size_t)y * pitch + x;
    run += *p;
    *p = run;
  }
}

// (row h + 1 / column w + 1 -- asked for by a +24 offset at a corner that rounds onto the border limit, out of bounds
//  upstream -- repeat the last row / column)
__device__ __forceinline__ int32_t smoothed(const int32_t* __restrict__ S, int w1, int h1, int px, int py, int dx, int dy) {
  constexpr int hk = BRIEF_KERNEL / 2;
  const int y = py + dy, x = px + dx;
  const int y1 = min(y + hk + 1, h1 - 1), x1 = min(x + hk + 1, w1 - 1);
  return S[(size_t)y1 * w1 + x1] - S[(size_t)y1 * w1 + x - hk] - S[(size_t)(y - hk) * w1 + x1] +
         S[(size_t)(y - hk) * w1 + x - hk];
}

struct ExtractCam {
  float fx, fy, cx, cy, cx_right, baseline, L[12], min_depth, max_depth;
  int identity_local, filter;
};

// one thread per (corner, descriptor byte)
__global__ void __launch_bounds__(256)
k_extract_points(const int32_t* __restrict__ S, int w, int h, const sf_keypoint* __restrict__ kpts,
                 const float* __restrict__ right_x, const uint8_t* __restrict__ status, int n, int bytes,
                 const int8_t* __restrict__ tests, ExtractCam cam, uint8_t* __restrict__ desc_tmp,
                 float* __restrict__ xyz_tmp, uint8_t* __restrict__ keep, ExtractBatch B) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int i = g / bytes, b = g - i * bytes;
  if (B.d_n) n = min(n, B.d_n[blockIdx.y]);
  if (i >= n) return;
  {
    const size_t o = (size_t)blockIdx.y * B.per_image;
    S += blockIdx.y * B.s_stride;
    kpts += o;
    if (right_x) right_x += o;
    if (status) status += o;
    desc_tmp += o * bytes; xyz_tmp += 3 * o; keep += o;
  }
  const sf_keypoint k = kpts[i];
  const bool inside = k.x >= (float)BRIEF_BORDER && k.x < (float)(w - BRIEF_BORDER) && k.y >= (float)BRIEF_BORDER &&
                      k.y < (float)(h - BRIEF_BORDER);
  if (inside) {
    const int px = (int)(k.x + 0.5f), py = (int)(k.y + 0.5f);
    unsigned v = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const char4 q = reinterpret_cast<const char4*>(tests)[8 * b + t];   // x1, y1, x2, y2
      v = (v << 1) | (unsigned)(smoothed(S, w + 1, h + 1, px, py, q.x, q.y) < smoothed(S, w + 1, h + 1, px, py, q.z, q.w));
    }
    desc_tmp[(size_t)i * bytes + b] = (uint8_t)v;
  }
  if (b != 0) return;
  const float qnan = __int_as_float(0x7FC00000);
  float p0 = qnan, p1 = qnan, p2 = qnan;
  if (inside && right_x && (!status || status[i])) {
    const float disparity = k.x - right_x[i];
    if (disparity != 0.0f && disparity > 0.0f && cam.baseline > 0.0f && cam.fx > 0.0f) {
      float c = 0.0f;
      if (cam.cx_right > 0.0f && cam.cx > 0.0f) c = cam.cx_right - cam.cx;
      const float W = cam.baseline / (disparity + c);
      const float x = (k.x - cam.cx) * W, y = (k.y - cam.cy) * W, z = cam.fx * W;
      if (isfinite(x) && isfinite(y) && isfinite(z) && (cam.min_depth < 0.0f || z > cam.min_depth) &&
          (cam.max_depth <= 0.0f || z <= cam.max_depth)) {
        if (cam.identity_local) {
          p0 = x; p1 = y; p2 = z;
        } else {
          p0 = ((cam.L[0] * x + cam.L[1] * y) + cam.L[2] * z) + cam.L[3];
          p1 = ((cam.L[4] * x + cam.L[5] * y) + cam.L[6] * z) + cam.L[7];
          p2 = ((cam.L[8] * x + cam.L[9] * y) + cam.L[10] * z) + cam.L[11];
        }
      }
    }
  }
  xyz_tmp[3 * i] = p0; xyz_tmp[3 * i + 1] = p1; xyz_tmp[3 * i + 2] = p2;
  keep[i] = (uint8_t)(inside && (!cam.filter || (isfinite(p0) && isfinite(p1) && isfinite(p2))));
}

// ONE workgroup: stable compaction into the store slot (and the optional wire copies)
__global__ void __launch_bounds__(256)
k_extract_commit(const sf_keypoint* __restrict__ kpts, const uint8_t* __restrict__ desc_tmp,
                 const float* __restrict__ xyz_tmp, const uint8_t* __restrict__ keep, int n, int bytes, int has3d,
                 uint32_t* __restrict__ st_desc, float* __restrict__ st_xyz, float4* __restrict__ st_kp,
                 int4* __restrict__ st_meta, int kcap, int w_dwords, int slot, uint8_t* __restrict__ desc_out,
                 float* __restrict__ xyz_out, sf_keypoint* __restrict__ kp_out, int32_t* __restrict__ rows_out,
                 ExtractBatch B) {
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (B.d_n) n = min(n, B.d_n[blockIdx.x]);
  {
    const size_t o = (size_t)blockIdx.x * B.per_image;
    slot += blockIdx.x;
    kpts += o; desc_tmp += o * bytes; xyz_tmp += 3 * o; keep += o;
    if (desc_out) desc_out += o * bytes;
    if (xyz_out) xyz_out += 3 * o;
    if (kp_out) kp_out += o;
    if (rows_out) rows_out += blockIdx.x;
  }
  uint8_t* d8 = reinterpret_cast<uint8_t*>(st_desc + (size_t)slot * kcap * w_dwords);
  float* dx = st_xyz + (size_t)slot * kcap * 3;
  float4* dk = st_kp + (size_t)slot * kcap;
  const int rowb = w_dwords * 4;
  int running = 0;
  for (int base = 0; base < n; base += 256) {
    const int i = base + tid;
    const bool f = i < n && keep[i];
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = wave_cnt[q];
      if (q < wave) before += c;
      total += c;
    }
    if (f) {
      const int o = running + before + __popcll(bal & ((1ull << lane) - 1ull));
      const sf_keypoint k = kpts[i];
      for (int b = 0; b < rowb; ++b) d8[(size_t)o * rowb + b] = b < bytes ? desc_tmp[(size_t)i * bytes + b] : (uint8_t)0;
      const float p0 = xyz_tmp[3 * i], p1 = xyz_tmp[3 * i + 1], p2 = xyz_tmp[3 * i + 2];
      dx[3 * o] = p0; dx[3 * o + 1] = p1; dx[3 * o + 2] = p2;
      int oc = k.octave & 255;
      oc = oc < 128 ? oc : (-128 | oc);
      dk[o] = make_float4(k.x, k.y, __int_as_float(oc), 0.f);
      if (desc_out) for (int b = 0; b < bytes; ++b) desc_out[(size_t)o * bytes + b] = desc_tmp[(size_t)i * bytes + b];
      if (xyz_out) { xyz_out[3 * o] = p0; xyz_out[3 * o + 1] = p1; xyz_out[3 * o + 2] = p2; }
      if (kp_out) kp_out[o] = k;
    }
    running += total;
    __syncthreads();
  }
  if (tid == 0) {
    st_meta[slot] = make_int4(running, has3d ? running : 0, running, bytes);
    if (rows_out) *rows_out = running;
  }
}

}  // namespace

// Default test set of a fresh handle: isotropic Gaussian offsets (sigma = patch / 5, the "G II" construction of the
// BRIEF paper), clipped to the patch, from a fixed linear congruential stream -- NOT OpenCV's table.
void sf_brief_default_pattern(int8_t* tests, int bytes) {
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto uni = [&]() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((s >> 11) + 1) / 9007199254740993.0;   // (0, 1)
  };
  for (int t = 0; t < 8 * bytes * 4; t += 2) {
    const double r = std::sqrt(-2.0 * std::log(uni())), a = 6.283185307179586 * uni();
    const double sigma = BRIEF_PATCH / 5.0;
    int x = (int)std::lround(sigma * r * std::cos(a)), y = (int)std::lround(sigma * r * std::sin(a));
    x = x < -BRIEF_PATCH / 2 ? -BRIEF_PATCH / 2 : (x > BRIEF_PATCH / 2 ? BRIEF_PATCH / 2 : x);
    y = y < -BRIEF_PATCH / 2 ? -BRIEF_PATCH / 2 : (y > BRIEF_PATCH / 2 ? BRIEF_PATCH / 2 : y);
    tests[t] = (int8_t)x;
    tests[t + 1] = (int8_t)y;
  }
}

// Launch sequence on the handle's stream; the store slot (kcap >= n, w dwords) has been reserved by the caller.
// n_img > 1: a batch -- image i at d_left + i * img_stride, its corners / right_x / status / optional copies at + i * n
// entries, its corner count in d_n[i] (device; `n` is then the per-image capacity), its store slot = slot + i.
int sf_launch_extract_batch(sf_context* c, const uint8_t* d_left, size_t img_stride, int n_img, int width, int height,
                            int pitch, const sf_keypoint* d_kpts, const float* d_right_x, const uint8_t* d_status, int n,
                            const int32_t* d_n, const sf_stereo_camera* cam, int bytes, const int8_t* d_tests,
                            uint32_t* st_desc, float* st_xyz, float4* st_kp, int4* st_meta, int kcap, int w_dwords,
                            int slot, uint8_t* d_desc_out, float* d_xyz_out, sf_keypoint* d_kpts_out,
                            int32_t* d_rows_out) {
  int rc;
  const size_t s_entries = (size_t)(width + 1) * (height + 1);
  const size_t rows_all = (size_t)std::max(n, 1) * n_img;
  if ((rc = sf_buf_reserve(c, c->ex_integral, s_entries * sizeof(int32_t) * n_img)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ex_desc, rows_all * bytes)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ex_xyz, rows_all * 12)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ex_keep, rows_all)) != SF_OK) return rc;
  int32_t* S = (int32_t*)c->ex_integral.p;
  ExtractBatch B;
  B.img_stride = img_stride; B.s_stride = s_entries; B.per_image = n; B.d_n = d_n;
  hipLaunchKernelGGL(k_integral_rows, dim3(height + 1, n_img), dim3(256), 0, c->stream, d_left, width, height, pitch, S, B);
  hipLaunchKernelGGL(k_integral_cols, dim3((width + 1 + 63) / 64, n_img), dim3(64), 0, c->stream, width, height, S, B);
  ExtractCam ec;
  ec.fx = cam->fx; ec.fy = cam->fy; ec.cx = cam->cx; ec.cy = cam->cy; ec.cx_right = cam->cx_right;
  ec.baseline = cam->baseline; ec.min_depth = cam->min_depth; ec.max_depth = cam->max_depth;
  ec.identity_local = 1;
  for (int e = 0; e < 12; ++e) {
    ec.L[e] = cam->local_transform[e];
    ec.identity_local = ec.identity_local && (ec.L[e] == ((e == 0 || e == 5 || e == 10) ? 1.0f : 0.0f));
  }
  ec.filter = cam->min_depth > 0.0f || cam->max_depth > 0.0f;
  if (n > 0) {
    const long long threads = (long long)n * bytes;
    hipLaunchKernelGGL(k_extract_points, dim3((unsigned)((threads + 255) / 256), n_img), dim3(256), 0, c->stream, S, width,
                       height, d_kpts, d_right_x, d_status, n, bytes, d_tests, ec, (uint8_t*)c->ex_desc.p,
                       (float*)c->ex_xyz.p, (uint8_t*)c->ex_keep.p, B);
  }
  hipLaunchKernelGGL(k_extract_commit, dim3(n_img), dim3(256), 0, c->stream, d_kpts, (const uint8_t*)c->ex_desc.p,
                     (const float*)c->ex_xyz.p, (const uint8_t*)c->ex_keep.p, n, bytes, d_right_x != nullptr, st_desc,
                     st_xyz, st_kp, st_meta, kcap, w_dwords, slot, d_desc_out, d_xyz_out, d_kpts_out, d_rows_out, B);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

int sf_launch_extract(sf_context* c, const uint8_t* d_left, int width, int height, int pitch, const sf_keypoint* d_kpts,
                      const float* d_right_x, const uint8_t* d_status, int n, const sf_stereo_camera* cam, int bytes,
                      const int8_t* d_tests, uint32_t* st_desc, float* st_xyz, float4* st_kp, int4* st_meta, int kcap,
                      int w_dwords, int slot, uint8_t* d_desc_out, float* d_xyz_out, sf_keypoint* d_kpts_out,
                      int32_t* d_rows_out) {
  return sf_launch_extract_batch(c, d_left, 0, 1, width, height, pitch, d_kpts, d_right_x, d_status, n, nullptr, cam, bytes,
                                 d_tests, st_desc, st_xyz, st_kp, st_meta, kcap, w_dwords, slot, d_desc_out, d_xyz_out,
                                 d_kpts_out, d_rows_out);
}
