// k_gftt.hip -- SURVEY.md section 8 row f3, the detector: cv::goodFeaturesToTrack as rtabmap's default feature type
// (GFTT/BRIEF) calls it from Feature2D::generateKeypoints (myRegistrationVis.cpp:281-283), on the device, so that a
// keyframe's left image can go from pixels to store slot without leaving HBM (k_extract.hip takes the corners from
// here; the right-image position of each corner -- rtabmap: pyramidal LK -- is still an input of that step).
//
// The algorithm and the ONE float summation order used where OpenCV's depends on its build are spelled out next to
// the CPU restatement the tests compare with (byte for byte; this file is compiled with -ffp-contract=off):
//   k_gftt_products    per pixel: Sobel dx, dy (scale 1/3060, BORDER_REFLECT_101), the three products
//   k_gftt_eig         per pixel: 3 x 3 sums of the products, eig = (a + c) - sqrt((a - c)^2 + b^2); max(eig)
//   k_gftt_candidates  interior pixels above quality * max that equal the maximum of their thresholded 3 x 3
//                      neighbourhood -> 64-bit keys (response bits << 32 | pixel index), one global atomic per workgroup
//   rocprim radix sort keys descending = decreasing response, ties by decreasing address (featureselect.cpp's
//                      greaterThanPtr); a plain library sort, the only library call of the product's compute path
//   k_gftt_select_lds  ONE wavefront walks the sorted list 64 candidates at a time: every lane tests its candidate
//                      against the corners already taken (a bitmap of the image in LDS), the 64 lanes settle
//                      conflicts among themselves in list order (in rounds, not in 64 turns), the survivors are
//                      appended -- the sequential minDistance rule of goodFeaturesToTrack, 64 candidates per step.
//   k_gftt_select      the same with the taken corners in per-cell lists in HBM: images whose bitmap does not fit
//                      LDS (above about 1.2 Mpixel), or SF_GFTT_LISTS=1.
#include <hip/hip_runtime.h>

#include <cstring>   // (rocprim's texture_cache_iterator.hpp calls memset without declaring it)

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "sf_internal.hpp"

namespace {

__device__ __forceinline__ int refl101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__global__ void __launch_bounds__(256)
k_gftt_products(const uint8_t* __restrict__ img, int w, int h, int pitch, float s1, float s2, float* __restrict__ dxx,
                float* __restrict__ dxy, float* __restrict__ dyy, size_t img_stride, size_t plane_stride) {
  // (blockIdx.z = image of a batch: images img_stride bytes apart, every image's planes plane_stride floats apart)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  img += blockIdx.z * img_stride;
  dxx += blockIdx.z * plane_stride; dxy += blockIdx.z * plane_stride; dyy += blockIdx.z * plane_stride;
  const int xm = refl101(x - 1, w), xp = refl101(x + 1, w);
  const uint8_t* ru = img + (size_t)refl101(y - 1, h) * pitch;
  const uint8_t* r0 = img + (size_t)y * pitch;
  const uint8_t* rd = img + (size_t)refl101(y + 1, h) * pitch;
  const float um = ru[xm], uc = ru[x], up = ru[xp], cm = r0[xm], cp = r0[xp], dm = rd[xm], dc = rd[x], dp = rd[xp];
  const float dx = s2 * (cp - cm) + s1 * ((up - um) + (dp - dm));
  const float cu = s2 * uc + s1 * (um + up);
  const float cd = s2 * dc + s1 * (dm + dp);
  const float dy = cd - cu;
  const size_t o = (size_t)y * w + x;
  dxx[o] = dx * dx; dxy[o] = dx * dy; dyy[o] = dy * dy;
}

__device__ __forceinline__ float box3(const float* __restrict__ p, int w, int yu, int y0, int yd, int xm, int x, int xp) {
  const float* a = p + (size_t)yu * w;
  const float* b = p + (size_t)y0 * w;
  const float* c = p + (size_t)yd * w;
  return (((a[xm] + a[x]) + a[xp]) + ((b[xm] + b[x]) + b[xp])) + ((c[xm] + c[x]) + c[xp]);
}

__global__ void __launch_bounds__(256)
k_gftt_eig(const float* __restrict__ dxx, const float* __restrict__ dxy, const float* __restrict__ dyy, int w, int h,
           float* __restrict__ eig, int* __restrict__ max_bits, size_t plane_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  dxx += blockIdx.z * plane_stride; dxy += blockIdx.z * plane_stride; dyy += blockIdx.z * plane_stride;
  eig += blockIdx.z * plane_stride;
  max_bits += blockIdx.z;
  float e = 0.f;
  if (x < w && y < h) {
    const int xm = refl101(x - 1, w), xp = refl101(x + 1, w), yu = refl101(y - 1, h), yd = refl101(y + 1, h);
    const float a = box3(dxx, w, yu, y, yd, xm, x, xp) * 0.5f, b = box3(dxy, w, yu, y, yd, xm, x, xp),
                c = box3(dyy, w, yu, y, yd, xm, x, xp) * 0.5f;
    e = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    eig[(size_t)y * w + x] = e;
  }
  // max(0, max e): positive floats order like their bit patterns
  int m = e > 0.f ? __float_as_int(e) : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off));
  // same-address atomics serialise in L2 (a few ns each): one per workgroup, and only when it can raise the maximum
  __shared__ int s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m > __builtin_nontemporal_load(max_bits)) atomicMax(max_bits, m);
  }
}

constexpr int GFTT_CAND_TILES = 4;      // 64 x 4 pixel tiles per workgroup of k_gftt_candidates
__global__ void __launch_bounds__(256)
k_gftt_candidates(const float* __restrict__ eig, int w, int h, const int* __restrict__ max_bits, double quality,
                  unsigned long long* __restrict__ keys, unsigned* __restrict__ count, unsigned cap, size_t plane_stride) {
  eig += blockIdx.z * plane_stride;
  max_bits += blockIdx.z;
  keys += (size_t)blockIdx.z * cap;
  count += blockIdx.z;
  // (the counter is ONE address: a returning atomic per 64 x 4 tile was most of this kernel's 20 us -- four tiles
  // per workgroup share one)
  __shared__ unsigned long long s_keys[256 * GFTT_CAND_TILES];
  __shared__ unsigned s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const float thr = (float)((double)__int_as_float(*max_bits) * quality);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
#pragma unroll
  for (int t = 0; t < GFTT_CAND_TILES; ++t) {
    const int y = (blockIdx.y * GFTT_CAND_TILES + t) * 4 + (threadIdx.x >> 6);
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
      const float v = eig[(size_t)y * w + x];
      if (v > thr) {
        float m = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const float u = eig[(size_t)(y + dy) * w + x + dx];
            const float tt = u > thr ? u : 0.f;
            m = tt > m ? tt : m;
          }
        if (v == m) {
          const unsigned p = atomicAdd(&s_n, 1u);
          s_keys[p] = ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)(y * w + x);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(count, s_n);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < s_n; i += 256)
    if (s_base + i < cap) keys[s_base + i] = s_keys[i];
}

__global__ void __launch_bounds__(64)
k_gftt_select(const unsigned long long* __restrict__ keys, int n, int w, int cell, int gw, int gh, float md2,
              int max_corners, int* __restrict__ head, int* __restrict__ next, int2* __restrict__ pts,
              sf_keypoint* __restrict__ kp_out, int cap, int32_t* __restrict__ n_out) {
  const int lane = threadIdx.x;
  int out = 0;
  for (int base = 0; base < n && (max_corners <= 0 || out < max_corners); base += 64) {
    const int i = base + lane;
    const bool valid = i < n;
    const int idx = valid ? (int)(unsigned)(keys[i] & 0xFFFFFFFFull) : 0;
    const int y = idx / w, x = idx - y * w;
    bool good = valid;
    if (good && cell > 0) {
      const int cx = x / cell, cy = y / cell;
      const int x1 = max(cx - 1, 0), y1 = max(cy - 1, 0), x2 = min(cx + 1, gw - 1), y2 = min(cy + 1, gh - 1);
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx = x1; xx <= x2 && good; ++xx)
          for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
            const int2 q = pts[j];
            const float dx = (float)(x - q.x), dy = (float)(y - q.y);
            if (dx * dx + dy * dy < md2) { good = false; break; }
          }
    }
    if (cell > 0) {
      // the 64 candidates of this step among themselves, in list order: lane j stands once every earlier lane
      // has been settled, and knocks out the later lanes within minDistance of it
      for (int j = 0; j < 63; ++j) {
        const int gj = __shfl((int)good, j), xj = __shfl(x, j), yj = __shfl(y, j);
        if (gj && lane > j && good) {
          const float dx = (float)(x - xj), dy = (float)(y - yj);
          if (dx * dx + dy * dy < md2) good = false;
        }
      }
    }
    const unsigned long long bal = __ballot(good);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    const int cnt = __popcll(bal);
    const int room = max_corners > 0 ? max_corners - out : 0x7FFFFFFF;
    if (good && rank < room) {
      const int o = out + rank;
      pts[o] = make_int2(x, y);
      if (cell > 0) next[o] = atomicExch(&head[(y / cell) * gw + x / cell], o);
      if (o < cap) {
        sf_keypoint k;
        k.x = (float)x; k.y = (float)y; k.size = 3.0f; k.angle = -1.0f; k.response = 0.0f; k.octave = 0; k.class_id = -1;
        kp_out[o] = k;
      }
    }
    out += min(cnt, room);
    __threadfence();          // the lists written above are read by every lane in the next step
  }
  if (lane == 0) *n_out = out;
}

// The same selection with the taken corners as a BITMAP of the image in LDS (one bit per pixel: 45 KB for 752 x 480):
// "no taken corner closer than minDistance" is a test of the bits inside the disc around the candidate, row by row,
// and nothing of a step waits for HBM (the cell lists of k_gftt_select cost a chain of dependent loads per step: 10 us
// per 64 candidates).  The distance test is the reference's -- dx * dx + dy * dy < minDistance^2 in float, exact for
// these integers -- only tabulated per row offset (span[dy] = largest |dx| inside the disc).  Within the 64 candidates
// of a step the lanes that are still standing take their turn in list order and knock out the later ones.
constexpr int GFTT_MAX_RADIUS = 64;
__global__ void __launch_bounds__(256)
k_gftt_select_lds(const unsigned long long* __restrict__ keys, int n, int w, int h, int wpr, float md2, int radius,
                  int max_corners, sf_keypoint* __restrict__ kp_out, int cap, int32_t* __restrict__ n_out,
                  const unsigned* __restrict__ d_count = nullptr, unsigned key_cap = 0) {
  extern __shared__ unsigned gf_bm[];
  if (d_count) {     // batch: one workgroup per image, the candidate count read on the device
    keys += (size_t)blockIdx.x * key_cap;
    n = (int)min(d_count[blockIdx.x], key_cap);
    kp_out += (size_t)blockIdx.x * cap;
    n_out += blockIdx.x;
  }
  __shared__ int span[2 * GFTT_MAX_RADIUS + 1];
  for (int i = threadIdx.x; i < wpr * h; i += 256) gf_bm[i] = 0u;
  if ((int)threadIdx.x <= 2 * radius) {
    const float dy = (float)((int)threadIdx.x - radius);
    int s = -1;
    while (s + 1 <= radius && (float)(s + 1) * (float)(s + 1) + dy * dy < md2) ++s;
    span[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  int out = 0;
  unsigned long long key = lane < n ? keys[lane] : 0ull;
  for (int base = 0; base < n && (max_corners <= 0 || out < max_corners); base += 64) {
    const int i = base + lane;
    const bool valid = i < n;
    const int idx = (int)(unsigned)(key & 0xFFFFFFFFull);
    key = i + 64 < n ? keys[i + 64] : 0ull;                  // (the next step's candidates, in flight during this one)
    const int y = idx / w, x = idx - y * w;
    bool good = valid;
    if (radius > 0 && good) {
      for (int dy = -radius; dy <= radius && good; ++dy) {
        const int yy = y + dy, s = span[dy + radius];
        if (yy < 0 || yy >= h || s < 0) continue;
        const int lo = max(x - s, 0), hi = min(x + s, w - 1);
        const unsigned* row = gf_bm + yy * wpr;
        for (int wd = lo >> 5; wd <= hi >> 5; ++wd) {
          unsigned m = 0xFFFFFFFFu;
          if (wd == lo >> 5) m &= 0xFFFFFFFFu << (lo & 31);
          if (wd == hi >> 5) m &= 0xFFFFFFFFu >> (31 - (hi & 31));
          if (row[wd] & m) good = false;
        }
      }
    }
    if (radius > 0) {
      // The candidates of this step among themselves, in list order: lane i stands unless a STANDING earlier lane is
      // within minDistance.  Every lane first collects the earlier lanes within minDistance (64 independent tests on
      // v_readlane broadcasts, no chain), then the standing set is settled in rounds: a lane whose earlier
      // neighbours are all settled is settled itself -- as many rounds as the longest chain of neighbours (2 - 4),
      // instead of 64 dependent turns.
      const unsigned long long alive = __ballot(good);
      unsigned c_lo = 0u, c_hi = 0u;
#pragma unroll
      for (int j = 0; j < 63; ++j) {
        const int xj = __builtin_amdgcn_readlane(x, j), yj = __builtin_amdgcn_readlane(y, j);
        const float dx = (float)(x - xj), dy = (float)(y - yj);
        const bool near = dx * dx + dy * dy < md2 && lane > j;
        if (j < 32) c_lo |= near ? (1u << j) : 0u; else c_hi |= near ? (1u << (j - 32)) : 0u;
      }
      const unsigned long long conf = (((unsigned long long)c_hi << 32) | c_lo) & alive;
      unsigned long long und = alive, acc = 0ull;
      while (und) {
        const bool mine = (und >> lane) & 1ull;
        const bool rej = mine && (conf & acc) != 0ull;
        const bool ok = mine && !rej && (conf & und) == 0ull;
        const unsigned long long r = __ballot(rej), a = __ballot(ok);
        acc |= a;
        und &= ~(r | a);
      }
      good = (acc >> lane) & 1ull;
    }
    const unsigned long long bal = __ballot(good);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    const int cnt = __popcll(bal);
    const int room = max_corners > 0 ? max_corners - out : 0x7FFFFFFF;
    if (good && rank < room) {
      const int o = out + rank;
      if (radius > 0) atomicOr(&gf_bm[y * wpr + (x >> 5)], 1u << (x & 31));
      if (o < cap) {
        sf_keypoint k;
        k.x = (float)x; k.y = (float)y; k.size = 3.0f; k.angle = -1.0f; k.response = 0.0f; k.octave = 0; k.class_id = -1;
        kp_out[o] = k;
      }
    }
    out += min(cnt, room);
  }
  if (lane == 0) *n_out = out;
}

}  // namespace

// Launch sequence on the handle's stream.  The candidate count crosses to the host once (the sort is sized by it).
int sf_launch_detect_corners(sf_context* c, const uint8_t* d_image, int width, int height, int pitch, int max_corners,
                             double quality_level, double min_distance, sf_keypoint* d_kpts_out, int cap,
                             int32_t* n_out) {
  const size_t np = (size_t)width * height;
  int rc;
  if ((rc = sf_buf_reserve(c, c->gf_planes, np * 4 * sizeof(float))) != SF_OK) return rc;
  const unsigned key_cap = (unsigned)np;
  if ((rc = sf_buf_reserve(c, c->gf_keys, (size_t)key_cap * 2 * sizeof(unsigned long long))) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->gf_scalar, 64)) != SF_OK) return rc;
  float* dxx = (float*)c->gf_planes.p;
  float* dxy = dxx + np;
  float* dyy = dxy + np;
  float* eig = dyy + np;
  unsigned long long* keys = (unsigned long long*)c->gf_keys.p;
  unsigned long long* keys_sorted = keys + key_cap;
  int* max_bits = (int*)c->gf_scalar.p;
  unsigned* count = (unsigned*)c->gf_scalar.p + 1;
  int32_t* d_n_out = (int32_t*)c->gf_scalar.p + 2;
  SF_HIP(c, hipMemsetAsync(c->gf_scalar.p, 0, 64, c->stream));
  const double scale = 1.0 / ((double)(1 << 2) * 3.0 * 255.0);
  const dim3 grid((width + 63) / 64, (height + 3) / 4), block(256);
  hipLaunchKernelGGL(k_gftt_products, grid, block, 0, c->stream, d_image, width, height, pitch, (float)(1.0 * scale),
                     (float)(2.0 * scale), dxx, dxy, dyy, (size_t)0, (size_t)0);
  hipLaunchKernelGGL(k_gftt_eig, grid, block, 0, c->stream, (const float*)dxx, (const float*)dxy, (const float*)dyy, width,
                     height, eig, max_bits, (size_t)0);
  const dim3 grid_c((width + 63) / 64, (height + 4 * GFTT_CAND_TILES - 1) / (4 * GFTT_CAND_TILES));
  hipLaunchKernelGGL(k_gftt_candidates, grid_c, block, 0, c->stream, (const float*)eig, width, height, (const int*)max_bits,
                     quality_level, keys, count, key_cap, (size_t)0);
  SF_HIP(c, hipGetLastError());
  unsigned h_count = 0;
  SF_HIP(c, hipMemcpyAsync(&h_count, count, 4, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  const int n = (int)std::min(h_count, key_cap);
  if (n > 0) {
    size_t tmp_bytes = 0;
    SF_HIP(c, rocprim::radix_sort_keys_desc(nullptr, tmp_bytes, keys, keys_sorted, (size_t)n, 0, 64, c->stream));
    if ((rc = sf_buf_reserve(c, c->gf_tmp, std::max<size_t>(tmp_bytes, 16))) != SF_OK) return rc;
    SF_HIP(c, rocprim::radix_sort_keys_desc(c->gf_tmp.p, tmp_bytes, keys, keys_sorted, (size_t)n, 0, 64, c->stream));
  }
  // taken corners as a bitmap in LDS when the image fits (up to about 1.2 Mpixel), cell lists in HBM otherwise
  const int wpr = (width + 31) / 32;
  const size_t bm_bytes = (size_t)wpr * height * sizeof(unsigned);
  const int radius = min_distance >= 1.0 ? (int)std::ceil(min_distance) : 0;
  if (bm_bytes <= 150 * 1024 && radius <= GFTT_MAX_RADIUS && !getenv("SF_GFTT_LISTS")) {
    if (!c->gf_select_attr) {
      SF_HIP(c, hipFuncSetAttribute((const void*)k_gftt_select_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      c->gf_select_attr = true;
    }
    hipLaunchKernelGGL(k_gftt_select_lds, dim3(1), dim3(256), bm_bytes, c->stream, (const unsigned long long*)keys_sorted, n,
                       width, height, wpr, (float)(min_distance * min_distance), radius, max_corners, d_kpts_out, cap, d_n_out);
  } else {
    const int cell = min_distance >= 1.0 ? (int)std::lrint(min_distance) : 0;
    const int gw = cell > 0 ? (width + cell - 1) / cell : 1, gh = cell > 0 ? (height + cell - 1) / cell : 1;
    const size_t list_n = (size_t)std::max(n, 1);
    const size_t n_int = (((size_t)gw * gh + list_n) + 1) & ~(size_t)1;       // (keeps the int2 array 8-byte aligned)
    if ((rc = sf_buf_reserve(c, c->gf_lists, n_int * sizeof(int) + list_n * sizeof(int2))) != SF_OK) return rc;
    int* head = (int*)c->gf_lists.p;
    int* next = head + (size_t)gw * gh;
    int2* pts = (int2*)(head + n_int);
    SF_HIP(c, hipMemsetAsync(head, 0xFF, (size_t)gw * gh * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_gftt_select, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)keys_sorted, n, width, cell,
                       gw, gh, (float)(min_distance * min_distance), max_corners, head, next, pts, d_kpts_out, cap, d_n_out);
  }
  SF_HIP(c, hipGetLastError());
  if (n_out) {
    SF_HIP(c, hipMemcpyAsync(n_out, d_n_out, 4, hipMemcpyDeviceToHost, c->stream));
    SF_HIP(c, hipStreamSynchronize(c->stream));
  }
  return SF_OK;
}


namespace {
__global__ void k_gftt_segments(const unsigned* __restrict__ count, unsigned key_cap, int n, unsigned* __restrict__ begin,
                                unsigned* __restrict__ end) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { begin[i] = (unsigned)i * key_cap; end[i] = (unsigned)i * key_cap + min(count[i], key_cap); }
}
}  // namespace

// The detector on a batch of images of one size, no host round trip: candidate counts stay on the device (a segmented
// sort takes the place of the sort sized by the host), one selection workgroup per image.  d_kpts_out [n_img][cap],
// d_n_out [n_img] (device).  Needs the LDS bitmap form of the selection (images up to about 1.2 Mpixel).
int sf_launch_detect_corners_batch(sf_context* c, const uint8_t* d_images, size_t img_stride, int n_img, int width, int height,
                                   int pitch, int max_corners, double quality_level, double min_distance,
                                   sf_keypoint* d_kpts_out, int cap, int32_t* d_n_out) {
  const size_t np = (size_t)width * height;
  const int wpr = (width + 31) / 32;
  const size_t bm_bytes = (size_t)wpr * height * sizeof(unsigned);
  const int radius = min_distance >= 1.0 ? (int)std::ceil(min_distance) : 0;
  if (bm_bytes > 150 * 1024 || radius > GFTT_MAX_RADIUS)
    return sf_fail(c, SF_ERANGE, "batched corner detection: %d x %d image does not fit the LDS selection bitmap", width, height);
  if (np * (size_t)n_img > 0xFFFFFFFFull) return sf_fail(c, SF_ERANGE, "batched corner detection: %d images of %zu pixels", n_img, np);
  int rc;
  if ((rc = sf_buf_reserve(c, c->gf_planes, np * 4 * sizeof(float) * n_img)) != SF_OK) return rc;
  const unsigned key_cap = (unsigned)np;
  if ((rc = sf_buf_reserve(c, c->gf_keys, (size_t)key_cap * 2 * sizeof(unsigned long long) * n_img)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->gf_scalar, 64 + (size_t)n_img * 16)) != SF_OK) return rc;
  float* dxx = (float*)c->gf_planes.p;                    // image i: planes at + i * 4 np (dxx, dxy, dyy, eig)
  float* dxy = dxx + np;
  float* dyy = dxy + np;
  float* eig = dyy + np;
  const size_t plane_stride = 4 * np;
  unsigned long long* keys = (unsigned long long*)c->gf_keys.p;
  unsigned long long* keys_sorted = keys + (size_t)key_cap * n_img;
  int* max_bits = (int*)((char*)c->gf_scalar.p + 64);
  unsigned* count = (unsigned*)(max_bits + n_img);
  unsigned* seg_begin = count + n_img;
  unsigned* seg_end = seg_begin + n_img;
  SF_HIP(c, hipMemsetAsync(max_bits, 0, (size_t)n_img * 8, c->stream));
  const double scale = 1.0 / ((double)(1 << 2) * 3.0 * 255.0);
  const dim3 grid((width + 63) / 64, (height + 3) / 4, n_img), block(256);
  hipLaunchKernelGGL(k_gftt_products, grid, block, 0, c->stream, d_images, width, height, pitch, (float)(1.0 * scale),
                     (float)(2.0 * scale), dxx, dxy, dyy, img_stride, plane_stride);
  hipLaunchKernelGGL(k_gftt_eig, grid, block, 0, c->stream, (const float*)dxx, (const float*)dxy, (const float*)dyy, width,
                     height, eig, max_bits, plane_stride);
  const dim3 grid_c((width + 63) / 64, (height + 4 * GFTT_CAND_TILES - 1) / (4 * GFTT_CAND_TILES), n_img);
  hipLaunchKernelGGL(k_gftt_candidates, grid_c, block, 0, c->stream, (const float*)eig, width, height, (const int*)max_bits,
                     quality_level, keys, count, key_cap, plane_stride);
  hipLaunchKernelGGL(k_gftt_segments, dim3((n_img + 63) / 64), dim3(64), 0, c->stream, (const unsigned*)count, key_cap, n_img,
                     seg_begin, seg_end);
  SF_HIP(c, hipGetLastError());
  size_t tmp_bytes = 0;
  SF_HIP(c, rocprim::segmented_radix_sort_keys_desc(nullptr, tmp_bytes, keys, keys_sorted, (unsigned)((size_t)key_cap * n_img),
                                                    (unsigned)n_img, seg_begin, seg_end, 0, 64, c->stream));
  if ((rc = sf_buf_reserve(c, c->gf_tmp, std::max<size_t>(tmp_bytes, 16))) != SF_OK) return rc;
  SF_HIP(c, rocprim::segmented_radix_sort_keys_desc(c->gf_tmp.p, tmp_bytes, keys, keys_sorted, (unsigned)((size_t)key_cap * n_img),
                                                    (unsigned)n_img, seg_begin, seg_end, 0, 64, c->stream));
  if (!c->gf_select_attr) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_gftt_select_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    c->gf_select_attr = true;
  }
  hipLaunchKernelGGL(k_gftt_select_lds, dim3(n_img), dim3(256), bm_bytes, c->stream, (const unsigned long long*)keys_sorted, 0,
                     width, height, wpr, (float)(min_distance * min_distance), radius, max_corners, d_kpts_out, cap, d_n_out,
                     (const unsigned*)count, key_cap);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
