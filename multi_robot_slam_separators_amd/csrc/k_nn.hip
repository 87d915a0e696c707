// k_nn.hip -- NetVLAD nearest-neighbour stage: DataHandler.find_matches
// (data_handler.py:166-209 of the reference), MI355X-native.
//
// The reference materialises the full N_l x N_r float64 distance matrix with scipy cdist, masks
// it, arg-sorts every row to take column 0, sorts the row minima and walks the first
// max_matches_nb rows.  Here:
//   k_nn_argmin   dense ||a||^2 + ||b||^2 - 2 a.b on the fp32 matrix cores
//                 (v_mfma_f32_32x32x2_f32: exact f32 fma chains), 128x128 tile per 256-thread
//                 workgroup, operands staged through LDS with a 36-float row pitch (conflict-free
//                 ds_read_b128), column masks folded into the column norms (+inf), ignored pairs
//                 checked against a per-row CSR list, and the row arg-min FUSED into the epilogue
//                 as a packed (dist bits << 32 | column) 64-bit minimum -- the N_l x N_r matrix
//                 never exists in HBM; only one partial minimum per (row, 64-column strip) does.
//   k_nn_select   reduces the partial minima per row, re-evaluates the winner's distance in
//                 float64 with the direct sqrt(sum (a-b)^2) form cdist uses, applies row masks.
//   host          sorts the N_l row minima (ties: lowest index) and performs the sequential walk
//                 of data_handler.py:193-205 (skip taken idx_other, accept under threshold, break
//                 at the first row over it).
#include <math.h>
#include <string.h>
#include <chrono>
#include <cmath>

#include <algorithm>
#include <vector>

#include "sf_internal.hpp"

namespace {

constexpr int NN_BM = 128, NN_BN = 128, NN_BK = 32, NN_PITCH = 36;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// XCD-aware tile rasterisation.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2
// each), so consecutive ids never share an L2.  Give every XCD a CONTIGUOUS chunk of the logical
// tile order (bijective for any grid size), and walk that order in bands of 8 row panels with the
// column index fastest inside a band: the ~100 workgroups an XCD runs concurrently then cover a
// compact 8 x 12 patch of tiles whose A/B panels are re-used out of that XCD's L2 instead of
// being re-fetched through the fabric (speed only; correctness never depends on placement).
__device__ __forceinline__ void nn_tile_of_block(int bid, int gx, int gy, int& tx, int& ty) {
  const int nwg = gx * gy;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int logical = xcd * q + (xcd < r ? xcd : r) + local;
  const int band = 8 * gx;
  const int g = logical / band, in = logical - g * band;
  const int first = g * 8;
  const int gsz = (gy - first) < 8 ? (gy - first) : 8;
  ty = first + in % gsz;
  tx = in / gsz;
}

// SF_NN_T256 = 0: the 128 x 128-tile filter for every non-128 level (A/B runs)
static bool nn_t256_on() {
  static const bool on = !(getenv("SF_NN_T256") && atoi(getenv("SF_NN_T256")) == 0);
  return on;
}

__global__ void __launch_bounds__(256)
k_nn_cast_rows(const double* __restrict__ src, float* __restrict__ dst, float* __restrict__ norms, int n, int dim,
               int ld) {
  // one wavefront per row: f64 -> f32 rows (zero padded to ld) + squared norms of the f32 values
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  double s = 0.0;   // squared norm of the float32 row, accumulated in float64 (error 2^-24 relative)
  for (int k = lane; k < ld; k += 64) {
    float v = (k < dim) ? (float)src[(size_t)row * dim + k] : 0.f;
    dst[(size_t)row * ld + k] = v;
    s += (double)v * (double)v;
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) norms[row] = (float)s;
}

__global__ void __launch_bounds__(256)
k_nn_copy_rows(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ norms, int n, int dim,
               int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  double s = 0.0;
  for (int k = lane; k < ld; k += 64) {
    float v = (k < dim) ? src[(size_t)row * dim + k] : 0.f;
    dst[(size_t)row * ld + k] = v;
    s += (double)v * (double)v;
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) norms[row] = (float)s;
}

// fp16 descriptors (BASELINE configs[4] ships NetVLAD in fp16): every half is exactly representable in fp32, so
// the database holds the same numbers and everything downstream is unchanged.
__global__ void __launch_bounds__(256)
k_nn_copy_rows_f16(const _Float16* __restrict__ src, float* __restrict__ dst, float* __restrict__ norms, int n,
                   int dim, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  double s = 0.0;
  for (int k = lane; k < ld; k += 64) {
    float v = (k < dim) ? (float)src[(size_t)row * dim + k] : 0.f;
    dst[(size_t)row * ld + k] = v;
    s += (double)v * (double)v;
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) norms[row] = (float)s;
}

// A: local rows [n_l_pad][ld], B: received rows [n_r_pad][ld] (both zero padded).
// nb_eff[j] = ||b_j||^2, or +inf for masked / padding columns.
// part[(strip) * n_l_pad + row] = packed min over the 64 columns of strip = blockIdx.x*2 + wave_col.
__global__ void __launch_bounds__(256)
k_nn_argmin(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ na,
            const float* __restrict__ nb_eff, const int* __restrict__ ign_ptr, const int* __restrict__ ign_col,
            unsigned long long* __restrict__ part, int n_l_pad, int ld, int gx, int gy) {
  __shared__ __attribute__((aligned(16))) float sA[NN_BM * NN_PITCH];
  __shared__ __attribute__((aligned(16))) float sB[NN_BN * NN_PITCH];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;           // wave tile origin: (64*wr, 64*wc)
  int tile_x, tile_y;
  nn_tile_of_block(blockIdx.x, gx, gy, tile_x, tile_y);
  const int row0 = tile_y * NN_BM, col0 = tile_x * NN_BN;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int srow = tid >> 3, sk4 = (tid & 7) * 4;   // staging: 32 rows x 8 float4 per pass
  for (int k0 = 0; k0 < ld; k0 += NN_BK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = srow + 32 * p;
      const float4 va = *reinterpret_cast<const float4*>(A + (size_t)(row0 + r) * ld + k0 + sk4);
      const float4 vb = *reinterpret_cast<const float4*>(B + (size_t)(col0 + r) * ld + k0 + sk4);
      *reinterpret_cast<float4*>(&sA[r * NN_PITCH + sk4]) = va;
      *reinterpret_cast<float4*>(&sB[r * NN_PITCH + sk4]) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const float4*>(&sA[(64 * wr + 32 * i + l31) * NN_PITCH + 8 * q + 4 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const float4*>(&sB[(64 * wc + 32 * j + l31) * NN_PITCH + 8 * q + 4 * h]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int strip = tile_x * 2 + wc;
  float nbj[2];
  int colj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    colj[j] = col0 + 64 * wc + 32 * j + l31;
    nbj[j] = nb_eff[colj[j]];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float nai = na[row];
      const int ip0 = ign_ptr[row], ip1 = ign_ptr[row + 1];
      // best and second-best key of this row over the strip's 64 columns (k_nn_select re-evaluates both in
      // float64 when they are closer than the error of this fp32 expansion)
      unsigned long long best = 0xFFFFFFFFFFFFFFFFull, second = 0xFFFFFFFFFFFFFFFFull;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float d2 = (nai + nbj[j]) - 2.f * acc[i][j][r];
        // -0 and tiny negative cancellation -> +0; NaN (a non-finite input) -> +inf: cdist reports NaN there
        // and numpy's argsort puts NaN last, so such a column can never be a row minimum
        d2 = d2 > 0.f ? d2 : (d2 == d2 ? 0.f : __int_as_float(0x7F800000));
        for (int e = ip0; e < ip1; ++e)
          if (ign_col[e] == colj[j]) d2 = __int_as_float(0x7F800000);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)colj[j];
        second = key < best ? best : (key < second ? key : second);
        best = key < best ? key : best;
      }
      // merge across the 32 lanes that hold this row (same lane>>5): the two smallest of the union
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        const unsigned long long ob = __shfl_xor(best, off), os = __shfl_xor(second, off);
        const unsigned long long hi = ob < best ? best : ob;         // the larger of the two bests
        const unsigned long long lo2 = os < second ? os : second;    // the smaller of the two seconds
        best = ob < best ? ob : best;
        second = hi < lo2 ? hi : lo2;
      }
      if (l31 == 0) {
        part[((size_t)strip * n_l_pad + row) * 2] = best;
        part[((size_t)strip * n_l_pad + row) * 2 + 1] = second;
      }
    }
  }
}

// Exact squared distance of two float32 rows, accumulated in float64 by one wavefront: lane l owns the
// float4 groups l, l + 64, ... (16-byte loads; the rows are zero padded to the pitch `ld`, a multiple of
// 32 floats, so the padding adds exact zeros).  Shared by k_nn_select and k_nn_refine: both precisions
// report bit-identical distances.
__device__ __forceinline__ double nn_exact_sq_dist(const float* __restrict__ a, const float* __restrict__ b, int ld,
                                                   int lane) {
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  double s = 0.0;
  for (int k = lane; k < ld / 4; k += 64) {
    const float4 x = a4[k], y = b4[k];
    const double d0 = (double)x.x - (double)y.x, d1 = (double)x.y - (double)y.y;
    const double d2 = (double)x.z - (double)y.z, d3 = (double)x.w - (double)y.w;
    s += d0 * d0;
    s += d1 * d1;
    s += d2 * d2;
    s += d3 * d3;
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

// One wavefront per local row: reduce the strips' keys, re-evaluate in float64, apply the row mask.
//
// The keys rank columns on the fp32 expansion |a|^2 + |b|^2 - 2 a.b, whose error is up to
// E = (ld + 8) 2^-24 (|a|^2 + max_j |b_j|^2) (ld-term fp32 dot product + the rounding of norms and sums), so a
// column whose true distance is the smallest may carry a key up to 2 E above the smallest key.  Every column
// that can be the true minimum is therefore re-evaluated exactly (float64, cdist's direct form) and the smallest
// exact distance wins, ties to the lowest column like a stable arg-min: a strip's best key when it lies within
// the band, and ALL 64 columns of a strip whose SECOND key lies within it (a third could hide behind it).
// With well-separated rows this is one evaluation per row, as before.
__global__ void __launch_bounds__(256)
k_nn_select(const unsigned long long* __restrict__ part, int n_strips, int n_l, int n_l_pad,
            const float* __restrict__ A, const float* __restrict__ B, int dim, int ld,
            const uint8_t* __restrict__ mask_local, const float* __restrict__ na, const float* __restrict__ nb_eff,
            const unsigned* __restrict__ nb_max_bits, const int* __restrict__ ign_ptr,
            const int* __restrict__ ign_col, int n_r, double* __restrict__ out_dist, int* __restrict__ out_idx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_l) return;
  unsigned long long best = 0xFFFFFFFFFFFFFFFFull;
  for (int s = lane; s < n_strips; s += 64) {
    const unsigned long long k = part[((size_t)s * n_l_pad + row) * 2];
    best = k < best ? k : best;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off);
    best = o < best ? o : best;
  }
  const unsigned best_bits = (unsigned)(best >> 32);
  const bool finite = (best_bits < 0x7F800000u) && !mask_local[row];
  double bd = (double)INFINITY;
  int bj = (int)(best & 0xFFFFFFFFu);
  if (finite) {
    const float E = (float)(ld + 8) * 5.9604645e-08f * (na[row] + __uint_as_float(*nb_max_bits));
    const float bound = __uint_as_float(best_bits) + 2.f * E;    // keys above it cannot be the true minimum
    const float* a = A + (size_t)row * ld;
    const int ip0 = ign_ptr[row], ip1 = ign_ptr[row + 1];
    bj = 0x7FFFFFFF;
    auto consider = [&](int j) {   // wave-uniform j
      const double d = nn_exact_sq_dist(a, B + (size_t)j * ld, ld, lane);
      if (d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
    };
    for (int s0 = 0; s0 < n_strips; s0 += 64) {
      const int s = s0 + lane;
      unsigned long long k1 = 0xFFFFFFFFFFFFFFFFull, k2 = 0xFFFFFFFFFFFFFFFFull;
      if (s < n_strips) {
        k1 = part[((size_t)s * n_l_pad + row) * 2];
        k2 = part[((size_t)s * n_l_pad + row) * 2 + 1];
      }
      const bool in1 = (unsigned)(k1 >> 32) < 0x7F800000u && __uint_as_float((unsigned)(k1 >> 32)) <= bound;
      const bool in2 = (unsigned)(k2 >> 32) < 0x7F800000u && __uint_as_float((unsigned)(k2 >> 32)) <= bound;
      unsigned long long m1 = __ballot(in1 && !in2), m2 = __ballot(in2);
      while (m1) {                                   // strips with ONE column in the band
        const int l = __ffsll((long long)m1) - 1;
        m1 &= m1 - 1;
        consider((int)(__shfl(k1, l) & 0xFFFFFFFFu));
      }
      while (m2) {                                   // strips with two (or possibly more): all 64 columns
        const int l = __ffsll((long long)m2) - 1;
        m2 &= m2 - 1;
        const int c0 = (s0 + l) * 64;
        for (int j = c0; j < c0 + 64 && j < n_r; ++j) {
          bool skip = !(nb_eff[j] < __int_as_float(0x7F800000));       // masked column
          for (int e = ip0; e < ip1 && !skip; ++e) skip = ign_col[e] == j;   // ignored pair
          if (!skip) consider(j);
        }
      }
    }
  }
  if (lane == 0) {
    const bool got = finite && bj != 0x7FFFFFFF;
    out_dist[row] = got ? sqrt(bd) : (double)INFINITY;
    // an all-inf row has no meaningful arg-min (numpy returns an unspecified index there)
    out_idx[row] = (best_bits < 0x7F800000u) ? (got ? bj : (int)(best & 0xFFFFFFFFu)) : 0;
  }
}


// ------------------------------------------------------------------------------------------------
// fp16 FILTER path (nn_precision = 1): the matrix cores only have to decide which (row, column)
// pairs CAN lie under netvlad_distance; every survivor is then re-evaluated exactly in float64.
// ------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) k_nn_maxabs(const float* __restrict__ rows, size_t n_elems, unsigned* out) {
  unsigned m = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(fabsf(rows[i])));
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// rows * scale -> fp16 (scale is a power of two: exact).  One wavefront per row; the fp16 row pitch
// ld16 (multiple of 64) may exceed the float pitch ld: the tail is zero filled.  Also emits the
// squared norm of the first `kprefix` elements of the float32 row (float64 accumulation).
__global__ void __launch_bounds__(256) k_nn_to_f16(const float* __restrict__ rows, _Float16* __restrict__ out,
                                                    float* __restrict__ prefix_norm, int n, int ld, int ld16,
                                                    int kprefix, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  double s = 0.0;
  for (int k = lane; k < ld16; k += 64) {
    const float v = (k < ld) ? rows[(size_t)row * ld + k] : 0.f;
    out[(size_t)row * ld16 + k] = (_Float16)(v * scale);
    if (k < kprefix) s += (double)v * (double)v;
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) prefix_norm[row] = (float)s;
}

// Filter inequality, rearranged so the epilogue costs one add, one fma and one compare per element:
//   na + nb - 2 acc/s < thr2 + 2 eps sqrt(na nb) + delta (na + nb)
//   <=>  acc > (A_i + B_j) - C_i * D_j
// with A_i = (na_i (1-delta) - thr2) s/2,  B_j = nb_j (1-delta) s/2,  C_i = eps s sqrt(na_i),  D_j = sqrt(nb_j)
// (delta = 2e-6 also covers the rounding of this rearrangement).  Masked / padding rows and columns get
// A = +inf (resp. B = +inf) and C = D = 0, so they can never be candidates.
// C_i and D_j are rounded UP to fp16-representable values (a larger error allowance keeps the filter a superset):
// the resident-panel kernel feeds them to the matrix cores as one more contraction step.  Their product is
// eps (s_a |a|)(s_b |b|); the scaled norms reach sqrt(k) 2^15, so the split is C = 64 eps s_a |a| (<= 2.4e4 at
// k = 128) and D = s_b |b| / 64 (<= 5.8e3): both inside fp16's range whatever the data.  delta = 2e-6 plus
// (k + 8) 2^-24 for the fp32 additions that now also carry -A_i (k_nn_filter_f16_k128 starts its accumulators there).
__device__ __forceinline__ float nn_up_to_f16(float x) {   // x >= 0
  _Float16 hv = (_Float16)x;
  if ((float)hv < x) {
    unsigned short b;
    __builtin_memcpy(&b, &hv, 2);
    ++b;                                   // next fp16 up (0x7BFF + 1 = +inf)
    __builtin_memcpy(&hv, &b, 2);
  }
  return (float)hv;
}
__global__ void k_nn_filter_row_coef(float2* rowc, const float* na, const uint8_t* mask_local, int n_l, int n_l_pad,
                                     float half_s, float thr2, float eps_sa64, float delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l_pad) return;
  const bool ok = i < n_l && !mask_local[i];
  const float n = ok ? na[i] : 0.f;
  rowc[i] = ok ? make_float2((n * (1.f - delta) - thr2) * half_s, nn_up_to_f16(eps_sa64 * sqrtf(n)))
               : make_float2(__int_as_float(0x7F800000), 0.f);
}
__global__ void k_nn_filter_col_coef(float2* colc, const float* nb, const uint8_t* mask_other, int n_r, int n_r_pad,
                                     float half_s, float sb_64, float delta) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_r_pad) return;
  const bool ok = j < n_r && !mask_other[j];
  const float n = ok ? nb[j] : 0.f;
  colc[j] = ok ? make_float2(n * (1.f - delta) * half_s, nn_up_to_f16(sb_64 * sqrtf(n)))
               : make_float2(__int_as_float(0x7F800000), 0.f);
}

// A16: local rows [n_l_pad][ld] fp16 (x scale_a), B16: received rows [n_r_pad][ld] fp16 (x scale_b).
// A (row, col) pair is emitted when its fp16 distance estimate can be below thr2 given the
// rigorous error bound  |dot16 - dot32| <= eps_rel * sqrt(na * nb).
__global__ void __launch_bounds__(256)
k_nn_filter_f16(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float2* __restrict__ rowc,
                const float2* __restrict__ colc, int ld, int kdims, int gx, int gy,
                uint2* __restrict__ cand, unsigned* __restrict__ cand_count, unsigned cand_cap) {
  __shared__ __attribute__((aligned(16))) float sA[NN_BM * NN_PITCH];   // 128 rows x 64 halfs (+ pad)
  __shared__ __attribute__((aligned(16))) float sB[NN_BN * NN_PITCH];
  __shared__ float2 sRowC[NN_BM], sColC[NN_BN];   // epilogue coefficients of this tile (no global loads there)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tile_x, tile_y;
  nn_tile_of_block(blockIdx.x, gx, gy, tile_x, tile_y);
  const int row0 = tile_y * NN_BM, col0 = tile_x * NN_BN;
  const int l31 = lane & 31, h = lane >> 5;
  const int ldw = ld / 2;                      // row pitch in dwords (ld halfs)
  if (tid < NN_BM) sRowC[tid] = rowc[row0 + tid];
  else sColC[tid - NN_BM] = colc[col0 + tid - NN_BM];   // visible after the K loop's barriers
  const float* Aw = reinterpret_cast<const float*>(A);
  const float* Bw = reinterpret_cast<const float*>(B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int srow = tid >> 3, sk4 = (tid & 7) * 4;   // staging: 32 rows x 8 x 16 B per pass
  // register-staged software pipeline: the global loads of K-tile t+1 are in flight while the
  // MFMAs of tile t run; they are written to LDS after the tile's reads are done.
  // (named registers, not arrays: hipcc demotes a float4[4] carried around the loop to scratch)
  const float* pa = Aw + (size_t)(row0 + srow) * ldw + sk4;
  const float* pb = Bw + (size_t)(col0 + srow) * ldw + sk4;
  const size_t rstep = (size_t)32 * ldw;
#define SF_LD4(ptr) (*reinterpret_cast<const float4*>(ptr))
  float4 ra0 = SF_LD4(pa), ra1 = SF_LD4(pa + rstep), ra2 = SF_LD4(pa + 2 * rstep), ra3 = SF_LD4(pa + 3 * rstep);
  float4 rb0 = SF_LD4(pb), rb1 = SF_LD4(pb + rstep), rb2 = SF_LD4(pb + 2 * rstep), rb3 = SF_LD4(pb + 3 * rstep);
  float* wa = &sA[srow * NN_PITCH + sk4];
  float* wb = &sB[srow * NN_PITCH + sk4];
  const int kdw = kdims / 2;                       // dwords of the (prefix of the) row to contract
  for (int k0 = 0; k0 < kdw; k0 += 32) {            // 32 dwords = 64 halfs per step
    __syncthreads();                                // previous tile's LDS reads are complete
    *reinterpret_cast<float4*>(wa) = ra0;
    *reinterpret_cast<float4*>(wa + 32 * NN_PITCH) = ra1;
    *reinterpret_cast<float4*>(wa + 64 * NN_PITCH) = ra2;
    *reinterpret_cast<float4*>(wa + 96 * NN_PITCH) = ra3;
    *reinterpret_cast<float4*>(wb) = rb0;
    *reinterpret_cast<float4*>(wb + 32 * NN_PITCH) = rb1;
    *reinterpret_cast<float4*>(wb + 64 * NN_PITCH) = rb2;
    *reinterpret_cast<float4*>(wb + 96 * NN_PITCH) = rb3;
    __syncthreads();
    {
      const int kn = (k0 + 32 < kdw) ? k0 + 32 : k0;   // last iteration re-reads its own tile (unused)
      ra0 = SF_LD4(pa + kn); ra1 = SF_LD4(pa + rstep + kn); ra2 = SF_LD4(pa + 2 * rstep + kn); ra3 = SF_LD4(pa + 3 * rstep + kn);
      rb0 = SF_LD4(pb + kn); rb1 = SF_LD4(pb + rstep + kn); rb2 = SF_LD4(pb + 2 * rstep + kn); rb3 = SF_LD4(pb + 3 * rstep + kn);
    }
#undef SF_LD4
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // 16 halfs of K per MFMA: lane holds k = 16 q + 8 h + 0..7
      half8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const half8*>(&sA[(64 * wr + 32 * i + l31) * NN_PITCH + 8 * q + 4 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const half8*>(&sB[(64 * wc + 32 * j + l31) * NN_PITCH + 8 * q + 4 * h]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  float2 cj[2];
  int colj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    colj[j] = 64 * wc + 32 * j + l31;
    cj[j] = sColC[colj[j]];
  }
  // both columns of a lane at once: packed fp32 add / fma (v_pk_add_f32, v_pk_fma_f32), 2 instead of 4
  // VALU operations per row; the N^2 compare epilogue is what bounds a short-prefix contraction
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 Bv = {cj[0].x, cj[1].x}, Dv = {cj[0].y, cj[1].y};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float2 ci = sRowC[row];
      const f32x2 Av = {ci.x, ci.x}, Cv = {-ci.y, -ci.y};
      const f32x2 rhs = __builtin_elementwise_fma(Cv, Dv, Av + Bv);
      const bool hit0 = acc[i][0][r] > rhs.x, hit1 = acc[i][1][r] > rhs.y;   // false for +inf / NaN (masked / padding)
      if (hit0 | hit1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 0 ? hit0 : hit1) {
            const unsigned pos = atomicAdd(cand_count, 1u);
            if (pos < cand_cap) cand[pos] = make_uint2((unsigned)(row0 + row), (unsigned)(col0 + colj[j]));
          }
        }
      }
    }
  }
}

// ---- the same filter on 256 x 256 tiles, LDS-DMA fed (round 5; the full-length contraction of --strict) ------------
// k_nn_filter_f16 above is the classic structure -- 128 x 128 tile, register-staged, two barriers per 64-deep K step --
// and sits at its ceiling (0.89 PF = 36 % of the fp16 peak on the 10 000 x 10 000 x 4096 contraction; fabric traffic
// 3 GB per launch: each operand panel is re-read 79 times).  This form: eight wavefronts (2 x 4) own a 256 x 256 tile,
// 128 x 64 of it each (8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16: 128 VGPRs); a K step of 64 is 64 KB of
// operands in LDS, double buffered (128 KB: one workgroup per CU); the NEXT step's operands arrive by
// global_load_lds_dwordx4 -- no staging registers, no ds_write -- issued a quarter per phase while the current step's
// four quadrants (16 MFMAs each) are contracted, and ONE barrier per K step (behind the issuing wavefronts'
// vmcnt(0)) publishes them.  The LDS image is [256 rows][8 x 16 B] per operand with the 16-byte column XORed by
// (row >> 1) & 7 (a linear image puts rows r and r + 2 on the same banks: 8-way conflicts on every fragment read); the
// DMA writes LDS linearly (wave-uniform base + lane x 16), so the permutation is applied to the SOURCE address and to the
// fragment reads (cdna_hip_programming.md section 5, rule 21).  Same inequality, same candidates as k_nn_filter_f16.
typedef __attribute__((address_space(3))) void* nn_lds_vp;
typedef __attribute__((address_space(1))) const void* nn_glb_vp;
#define NN256_LDS (2 * 2 * 256 * 128)
typedef float nn_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512, 2)
k_nn_filter_f16_t256(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float2* __restrict__ rowc,
                     const float2* __restrict__ colc, int ld, int kdims, int gx, int gy, int n_l_pad, int n_r_pad,
                     uint2* __restrict__ cand, unsigned* __restrict__ cand_count, unsigned cand_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nn256_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  int tile_x, tile_y;
  nn_tile_of_block(blockIdx.x, gx, gy, tile_x, tile_y);
  const int row0 = tile_y * 256, col0 = tile_x * 256;

  // ---- DMA source addresses: piece (op, half, j) of a K step = 64 lanes x 16 B = 8 rows x 128 B of the LDS image -------
  // slot s = j * 512 + tid of the half-tile: image row s >> 3, image column s & 7 <- source column (s & 7) ^ ((row >> 1) & 7)
  const char* srcA[2][2];
  const char* srcB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int sl = j * 512 + tid;
      const int r = sl >> 3, cs = (sl & 7) ^ ((r >> 1) & 7);
      const int ra = min(row0 + h * 128 + r, n_l_pad - 1), rb = min(col0 + h * 128 + r, n_r_pad - 1);   // (rows past the
      srcA[h][j] = reinterpret_cast<const char*>(A + (size_t)ra * ld) + cs * 16;                          //  padding: masked below)
      srcB[h][j] = reinterpret_cast<const char*>(B + (size_t)rb * ld) + cs * 16;
    }
  const int dma_dst = wave * 1024;                                     // + op * 32768 + h * 16384 + j * 8192 (+ buffer)
  auto issue = [&](int buf, int k0, int quarter) {                     // quarter: 0 = A half 0, 1 = A half 1, 2 = B half 0, 3 = B half 1
    const int op = quarter >> 1, h = quarter & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* src = (op ? srcB[h][j] : srcA[h][j]) + (size_t)k0 * 2;
      __builtin_amdgcn_global_load_lds((nn_glb_vp)src,
                                       (nn_lds_vp)(nn256_lds + buf * 65536 + op * 32768 + h * 16384 + j * 8192 + dma_dst), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside an operand's 32 KB image) ----------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int swz = (fr >> 1) & 7;
  const int offA = (wr * 128 + fr) * 128, offB = (wc * 64 + fr) * 128;
  const int col_k0 = ((0 + fq) ^ swz) * 16, col_k1 = ((4 + fq) ^ swz) * 16;

  nn_f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = nn_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = kdims / 64;
#pragma unroll
  for (int qtr = 0; qtr < 4; ++qtr) issue(0, 0, qtr);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const unsigned char* bA = nn256_lds + cur * 65536;
    const unsigned char* bB = bA + 32768;
    const bool more = t + 1 < nk;
    half8 a[4][2], b[2][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      // quadrants in the order (m half, n half) = (0,0) (0,1) (1,1) (1,0): one operand's fragments change per phase
      const int mh = ph >> 1, nh = (ph == 1 || ph == 2) ? 1 : 0;
      if (more) issue(cur ^ 1, (t + 1) * 64, ph);
      if (ph == 0 || ph == 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const unsigned char* p = bA + offA + (mh * 4 + m) * 2048;
          a[m][0] = *reinterpret_cast<const half8*>(p + col_k0);
          a[m][1] = *reinterpret_cast<const half8*>(p + col_k1);
        }
      }
      if (ph != 2) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const unsigned char* p = bB + offB + (nh * 2 + n) * 2048;
          b[n][0] = *reinterpret_cast<const half8*>(p + col_k0);
          b[n][1] = *reinterpret_cast<const half8*>(p + col_k1);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[mh * 4 + m][nh * 2 + n] =
                __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][ks], b[n][ks], acc[mh * 4 + m][nh * 2 + n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's pieces of step t + 1 have landed ...
    __builtin_amdgcn_s_barrier();                         // ... everyone's have, and everyone is done reading step t
  }

  // ---- epilogue: the filter inequality per element (coefficients of the tile staged in LDS) ------------------------------
  float2* sRowC = reinterpret_cast<float2*>(nn256_lds);
  float2* sColC = sRowC + 256;
  if (tid < 256) sRowC[tid] = (row0 + tid < n_l_pad) ? rowc[row0 + tid] : make_float2(__int_as_float(0x7F800000), 0.f);
  else sColC[tid - 256] = (col0 + tid - 256 < n_r_pad) ? colc[col0 + tid - 256] : make_float2(__int_as_float(0x7F800000), 0.f);
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int col = wc * 64 + n * 16 + fr;
    const float2 cj = sColC[col];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wr * 128 + m * 16 + fq * 4 + j;
        const float2 ci = sRowC[row];
        const float rhs = __builtin_fmaf(-ci.y, cj.y, ci.x + cj.x);
        if (acc[m][n][j] > rhs) {      // false for +inf / NaN (masked / padding)
          const unsigned pos = atomicAdd(cand_count, 1u);
          if (pos < cand_cap) cand[pos] = make_uint2((unsigned)(row0 + row), (unsigned)(col0 + col));
        }
      }
    }
  }
}

// ---- the same filter for a 128-dimension contraction, with the row panel resident ----------------------------
// k_nn_filter_f16 at k = 128 spent its time outside the matrix cores: every 128 x 128 tile paid the global-load
// latency of both operand panels (two barriers per 64-dimension step, nothing to overlap it with) and ~10 VALU
// instructions per pair of outputs in the compare epilogue (profiles/r02b: 73 us for 25.6 GFLOP).  Here
//   * a workgroup keeps ONE 128-row panel of the local descriptors in LDS (32 KB) and walks a strip of column
//     tiles: the received panel of tile t + 1 is in flight (registers) while tile t is contracted, one LDS buffer,
//     two barriers per TILE;
//   * the inequality  acc > (A_i + B_j) - C_i D_j  is folded into the contraction: the accumulators START at -A_i
//     (the MFMA's C operand, free), C_i D_j is a ninth contraction step (C_i and D_j are fp16-representable by
//     construction, their product is exact), and what is left per output is  acc' > B_j;
//   * that compare is done on the MAXIMUM of each lane's 16 outputs of a 32 x 32 block (v_max3), one compare and a
//     ballot per block; only blocks with a hit look at their 16 outputs one by one;
//   * hits go to a list in LDS and the workgroup takes its slots of the global candidate list with ONE atomic at the
//     end of its strip: one returning atomic per hit on the single global counter serialises in L2 at ~3 ns each,
//     which -- not the contraction -- was what both this kernel and k_nn_filter_f16 took 73 us for at ~25 000 hits.
// Per 64 x 64 wavefront tile: 36 MFMAs (1152 matrix-pipe cycles) against ~150 other instructions.
constexpr int NN_KP = 68;                 // LDS row pitch in dwords: 64 dwords (128 halfs) + 4
constexpr int NN_K128_HITS = 1024;        // candidates a workgroup collects in LDS before it takes global slots
constexpr int NN_K128_LDS = 2 * NN_BM * NN_KP * 4 + NN_K128_HITS * 8 + 16;

__global__ void __launch_bounds__(256, 2)
k_nn_filter_f16_k128(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float2* __restrict__ rowc,
                     const float2* __restrict__ colc, int gx, int gy, int tiles_per_strip,
                     uint2* __restrict__ cand, unsigned* __restrict__ cand_count, unsigned cand_cap,
                     unsigned* __restrict__ next_count) {
  extern __shared__ __attribute__((aligned(16))) float nn_lds[];
  // the NEXT launch's counter block (the launches alternate between two): zeroed here, so that no memset launch
  // stands in front of the next query's filter
  if (blockIdx.x == 0 && threadIdx.x < 16 && next_count) next_count[threadIdx.x] = 0u;
  float* sA = nn_lds;                         // [128][NN_KP]
  float* sB = nn_lds + NN_BM * NN_KP;         // [128][NN_KP]
  uint2* s_hits = reinterpret_cast<uint2*>(nn_lds + 2 * NN_BM * NN_KP);   // [NN_K128_HITS]
  unsigned* s_nhits = reinterpret_cast<unsigned*>(s_hits + NN_K128_HITS);  // [0] count, [1] global base
  const int tid = threadIdx.x;
  if (tid == 0) s_nhits[0] = 0;               // (visible after the first barrier of the tile loop)
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int strips = (gx + tiles_per_strip - 1) / tiles_per_strip;
  const int tile_y = blockIdx.x / strips, strip = blockIdx.x - tile_y * strips;   // (gy row panels) x strips
  const int t_first = strip * tiles_per_strip;
  const int t_last = min(gx, t_first + tiles_per_strip);
  const int row0 = tile_y * NN_BM;
  const float4* Ag = reinterpret_cast<const float4*>(A) + (size_t)row0 * 16;   // 16 float4 per 128-half row

  // the column panel of the first tile is requested first, then the row panel (written to LDS straight away)
  // (named registers, not an array: hipcc demotes a float4[8] carried around the tile loop to scratch)
  float4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
#define SF_K128_LOAD(Bg)                                                                              \
  rb0 = (Bg)[tid]; rb1 = (Bg)[tid + 256]; rb2 = (Bg)[tid + 512]; rb3 = (Bg)[tid + 768];               \
  rb4 = (Bg)[tid + 1024]; rb5 = (Bg)[tid + 1280]; rb6 = (Bg)[tid + 1536]; rb7 = (Bg)[tid + 1792]
#define SF_K128_STORE(u, v)                                                                           \
  *reinterpret_cast<float4*>(&sB[((tid + 256 * (u)) >> 4) * NN_KP + ((tid + 256 * (u)) & 15) * 4]) = (v)
  {
    const float4* Bg = reinterpret_cast<const float4*>(B) + (size_t)t_first * NN_BN * 16;
    SF_K128_LOAD(Bg);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int ch = tid + 256 * u;
    *reinterpret_cast<float4*>(&sA[(ch >> 4) * NN_KP + (ch & 15) * 4]) = Ag[ch];
  }
  // accumulator start values -A_i and the row half of the ninth step, for this lane's rows
  f32x16 cinit[2];
  half8 a_ext[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[i][r] = -rowc[row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h].x;
    const float ci = rowc[row0 + 64 * wr + 32 * i + l31].y;
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    a_ext[i] = z;
    if (h == 0) a_ext[i][0] = (_Float16)ci;      // k = 0 of the extra step lives in lanes 0..31
  }

  for (int t = t_first; t < t_last; ++t) {
    const int col0 = t * NN_BN;
    __syncthreads();                              // the previous tile's reads of sB are complete
    SF_K128_STORE(0, rb0); SF_K128_STORE(1, rb1); SF_K128_STORE(2, rb2); SF_K128_STORE(3, rb3);
    SF_K128_STORE(4, rb4); SF_K128_STORE(5, rb5); SF_K128_STORE(6, rb6); SF_K128_STORE(7, rb7);
    __syncthreads();
    if (t + 1 < t_last) {
      const float4* Bg = reinterpret_cast<const float4*>(B) + (size_t)(t + 1) * NN_BN * 16;
      SF_K128_LOAD(Bg);
    }
    float bj[2];
    half8 b_ext[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float2 cj = colc[col0 + 64 * wc + 32 * j + l31];
      bj[j] = cj.x;
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      b_ext[j] = z;
      if (h == 0) b_ext[j][0] = (_Float16)cj.y;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // 16 halfs of K per MFMA: lane holds k = 16 q + 8 h + 0..7
      half8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const half8*>(&sA[(64 * wr + 32 * i + l31) * NN_KP + 8 * q + 4 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const half8*>(&sB[(64 * wc + 32 * j + l31) * NN_KP + 8 * q + 4 * h]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], q == 0 ? cinit[i] : acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_ext[i], b_ext[j], acc[i][j], 0, 0, 0);

#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x16& v = acc[i][j];
        // the block test runs on the BIT PATTERNS as signed integers (v_max3_i32; a float maximum of MFMA results
        // costs a canonicalising instruction per operand): B_j >= +0, among non-negative floats the integer order is
        // the float order, and every negative float is a negative integer.  A NaN output (NaN descriptor) can pass
        // this test; the per-output float compare below then rejects it.
        int m = max(max(__float_as_int(v[0]), __float_as_int(v[1])), __float_as_int(v[2]));
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = max(max(m, __float_as_int(v[r])), __float_as_int(v[r + 1]));
        m = max(m, __float_as_int(v[15]));
        if (__ballot(m > __float_as_int(bj[j])) != 0ull) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (v[r] > bj[j]) {
              const uint2 e = make_uint2((unsigned)(row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h),
                                         (unsigned)(col0 + 64 * wc + 32 * j + l31));
              const unsigned lp = atomicAdd(&s_nhits[0], 1u);
              if (lp < (unsigned)NN_K128_HITS) {
                s_hits[lp] = e;
              } else {                        // dense tile: past the LDS list every hit takes its own slot
                const unsigned pos = atomicAdd(cand_count, 1u);
                if (pos < cand_cap) cand[pos] = e;
              }
            }
          }
        }
      }
    }
  }
#undef SF_K128_LOAD
#undef SF_K128_STORE
  __syncthreads();
  const unsigned nh = min(s_nhits[0], (unsigned)NN_K128_HITS);
  if (tid == 0 && nh) s_nhits[1] = atomicAdd(cand_count, nh);
  __syncthreads();
  if (nh) {
    const unsigned base = s_nhits[1];
    for (unsigned e = tid; e < nh; e += 256)
      if (base + e < cand_cap) cand[base + e] = s_hits[e];
  }
}

// ---- round 3: the row panel in REGISTERS, the column tiles by LDS-DMA -----------------------------------------
// k_nn_filter_f16_k128 above reads both operands of every MFMA from LDS (32 ds_read_b128 per wavefront and tile for
// 36 MFMAs), stages the column tile through 32 registers per lane and fills its single buffer between two barriers.
// Here a wavefront's share of the row panel (64 rows x 128 halfs = 64 VGPRs per lane) is loaded once per strip
// straight from global memory in MFMA operand layout and kept; LDS holds only column tiles, two of them, filled by
// global_load_lds_dwordx4 (no staging registers, no ds_write): tile t + 1 is in flight while tile t is contracted,
// ONE barrier per tile, 16 ds_read_b128 per wavefront and tile.  An LDS-DMA instruction writes 1 KiB contiguously
// (four 256-byte rows), so the image cannot be padded; it is XOR-swizzled instead -- the 16-byte chunk c of row r
// sits at chunk c ^ (r & 15), applied on the SOURCE address of the DMA and on the ds_read address -- which makes every
// ds_read_b128 lane group (16 rows, one chunk each) hit 16 different bank quads.
// Same arithmetic, same order of the contraction steps, same candidate set as the kernel above (a candidate's
// position in the list is free: the consumers take row minima).
constexpr int NN_K128R_ROWS = NN_BN * 64;            // dwords of the rows of one column tile (128 rows x 256 B)
constexpr int NN_K128R_TILE = NN_K128R_ROWS + 256;   // + the tile's 128 column coefficients (one more 1 KiB piece)

template <int ABL>     // ABL != 0: timing-only ablations (tools/nn_filter_time.py), never launched by the product path
__global__ void __launch_bounds__(256, 2)
k_nn_filter_f16_k128r(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float2* __restrict__ rowc,
                      const float2* __restrict__ colc, int gx, int gy, int tiles_per_strip,
                      uint2* __restrict__ cand, unsigned* __restrict__ cand_count, unsigned cand_cap,
                      unsigned* __restrict__ next_count) {
  // The two tile buffers are two OBJECTS, and the tile loop below is unrolled by two so that every access names its
  // buffer statically: the compiler orders a ds_read behind an outstanding LDS-DMA (s_waitcnt vmcnt(0)) whenever it cannot
  // prove that the two do not alias -- with one array and a runtime buffer index it waited for the next tile's DMA
  // right after issuing it, in front of the current tile's first operand read (no overlap at all).
  __shared__ __attribute__((aligned(16))) float sB0[NN_K128R_TILE];   // [128 rows][16 chunks of 16 B], swizzled, + coefficients
  __shared__ __attribute__((aligned(16))) float sB1[NN_K128R_TILE];
  __shared__ uint2 s_hits[NN_K128_HITS];
  __shared__ unsigned s_nhits[2];                                      // [0] count, [1] global base
  if (blockIdx.x == 0 && threadIdx.x < 16 && next_count) next_count[threadIdx.x] = 0u;
  const int tid = threadIdx.x;
  if (tid == 0) s_nhits[0] = 0;               // (visible after the first barrier)
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int strips = (gx + tiles_per_strip - 1) / tiles_per_strip;
  const int tile_y = blockIdx.x / strips, strip = blockIdx.x - tile_y * strips;   // (gy row panels) x strips
  const int t_first = strip * tiles_per_strip;
  const int t_last = min(gx, t_first + tiles_per_strip);
  const int row0 = tile_y * NN_BM;

  // LDS-DMA of one 128-row block: 32 pieces of 1 KiB (4 rows), 8 per wavefront; lane l of piece p fills row
  // 4 p + (l >> 4), physical chunk l & 15, i.e. fetches logical chunk (l & 15) ^ (row & 15)
  // (pieces u and u + 4 of a wavefront are 16 rows apart: same swizzle, 4096 bytes further)
  unsigned dma_off[4];                       // byte offset of this lane's source inside a block, per piece & 3
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = 4 * (8 * wave + u) + (lane >> 4);
    dma_off[u] = (unsigned)(row * 256 + (((lane & 15) ^ (row & 15)) << 4));
  }
  // one 128-row block (rows at `rows_g`, 256 B each; its 128 coefficient pairs at `coef_g`) -> LDS buffer `buf`
  auto dma_block = [&](const void* rows_g, const float2* coef_g, float* buf) {
    const char* Bg = reinterpret_cast<const char*>(rows_g);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      __builtin_amdgcn_global_load_lds((nn_glb_vp)(Bg + (u >> 2) * 4096 + dma_off[u & 3]),
                                       (nn_lds_vp)(buf + (8 * wave + u) * 256), 16, 0, 0);
    // the block's coefficients ride along (an ordinary load in the tile loop would make the compiler wait for
    // ALL outstanding vector-memory operations, the DMA included, at its first use)
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((nn_glb_vp)(reinterpret_cast<const char*>(coef_g) + lane * 16),
                                       (nn_lds_vp)(buf + NN_K128R_ROWS), 16, 0, 0);
  };
  // prologue: the first column tile into buffer 0 and the ROW panel, with its coefficients, into buffer 1 -- by the
  // same coalesced 1 KiB pieces; fragment-shaped loads straight from global memory (32 rows x 32 B per instruction)
  // and 32 scalar coefficient loads per lane made the prologue a third of the kernel at 10 000 x 10 000
  if (!(ABL & 16)) {
    dma_block(B + (size_t)t_first * NN_BN * 128, colc + (size_t)t_first * NN_BN, sB0);
    dma_block(A + (size_t)row0 * 128, rowc + row0, sB1);
  }
  __syncthreads();                                    // (vmcnt(0) in front of it: both blocks have landed)
  // this wavefront's rows of the panel, in operand layout: lane (l31, h) holds halfs 16 q + 8 h .. + 7 of row
  // 64 wr + 32 i + l31 for q = 0 .. 7
  half8 a[2][8];
  f32x16 cinit[2];
  _Float16 ci16[2];                                // the row half of the ninth step (k = 0 lives in lanes 0..31)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 64 * wr + 32 * i + l31;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      a[i][q] = *reinterpret_cast<const half8*>(&sB1[row * 64 + (((2 * q + h) ^ (l31 & 15)) << 2)]);
    // accumulator start values -A_i of this lane's 16 rows: four groups of four consecutive rows
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4* cp = reinterpret_cast<const float4*>(&sB1[NN_K128R_ROWS + 2 * (64 * wr + 32 * i + 8 * g + 4 * h)]);
      const float4 c01 = cp[0], c23 = cp[1];
      cinit[i][4 * g] = -c01.x; cinit[i][4 * g + 1] = -c01.z; cinit[i][4 * g + 2] = -c23.x; cinit[i][4 * g + 3] = -c23.z;
    }
    const float cy = sB1[NN_K128R_ROWS + 2 * row + 1];
    ci16[i] = h == 0 ? (_Float16)cy : (_Float16)0.f;
  }
  // ds_read address of this lane inside a tile: row 64 wc + 32 j + l31 (row & 15 = l31 & 15), chunk (2 q + h) ^ (l31 & 15)
  const int rd_row = (64 * wc + l31) * 64;            // dwords
  const int rd_sw = l31 & 15;
  __syncthreads();                                    // every wavefront has its panel rows: buffer 1 is free

  // one column tile out of `cur`
  auto contract = [&](const float* cur, int t) {
    const int col0 = t * NN_BN;
    // the two 32-column halves of the wavefront's tile one after the other: 32 accumulator registers live, not 64
    // (two independent chains of dependent MFMAs keep the pipe full)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16 acc[2];
#pragma unroll
      for (int q = 0; q < 8; ++q) {   // 16 halfs of K per MFMA: lane holds k = 16 q + 8 h + 0..7
        const half8 b = (ABL & 2) ? a[1][q ^ 1]
                                  : *reinterpret_cast<const half8*>(&cur[rd_row + 32 * 64 * j + (((2 * q + h) ^ rd_sw) << 2)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][q], b, q == 0 ? cinit[i] : acc[i], 0, 0, 0);
      }
      const float2 cj = *reinterpret_cast<const float2*>(&cur[NN_K128R_ROWS + 2 * (64 * wc + 32 * j + l31)]);
      const float bj = cj.x;
      {
        half8 b_ext = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) b_ext[0] = (_Float16)cj.y;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          half8 a_ext = {0, 0, 0, 0, 0, 0, 0, 0};
          a_ext[0] = ci16[i];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_ext, b_ext, acc[i], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x16& v = acc[i];
        // (block test on the bit patterns as signed integers: see k_nn_filter_f16_k128)
        int m = max(max(__float_as_int(v[0]), __float_as_int(v[1])), __float_as_int(v[2]));
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = max(max(m, __float_as_int(v[r])), __float_as_int(v[r + 1]));
        m = max(m, __float_as_int(v[15]));
        if (ABL & 1) { asm volatile("" :: "v"(m)); continue; }
        if (__ballot(m > __float_as_int(bj)) != 0ull) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (v[r] > bj) {
              const uint2 e = make_uint2((unsigned)(row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h),
                                         (unsigned)(col0 + 64 * wc + 32 * j + l31));
              const unsigned lp = atomicAdd(&s_nhits[0], 1u);
              if (lp < (unsigned)NN_K128_HITS) {
                s_hits[lp] = e;
              } else {                        // dense tile: past the LDS list every hit takes its own slot
                const unsigned pos = atomicAdd(cand_count, 1u);
                if (pos < cand_cap) cand[pos] = e;
              }
            }
          }
        }
      }
    }
  };
  auto dma_tile = [&](int t, float* buf) {
    if (!(ABL & 4)) dma_block(B + (size_t)t * NN_BN * 128, colc + (size_t)t * NN_BN, buf);
  };
  for (int t = t_first; t < ((ABL & 8) ? t_first : t_last); t += 2) {
    if (t + 1 < t_last) dma_tile(t + 1, sB1);       // (its last reads ended before the previous barrier)
    contract(sB0, t);
    __syncthreads();            // vmcnt(0) + barrier: the next tile has landed, this tile's reads are complete
    if (t + 1 >= t_last) break;
    if (t + 2 < t_last) dma_tile(t + 2, sB0);
    contract(sB1, t + 1);
    __syncthreads();
  }
  const unsigned nh = min(s_nhits[0], (unsigned)NN_K128_HITS);
  if (tid == 0 && nh) s_nhits[1] = atomicAdd(cand_count, nh);
  __syncthreads();
  if (nh) {
    const unsigned base = s_nhits[1];
    for (unsigned e = tid; e < nh; e += 256)
      if (base + e < cand_cap) cand[base + e] = s_hits[e];
  }
}

// one wavefront per candidate: exact float64 distance in cdist's direct form
__global__ void __launch_bounds__(256)
k_nn_refine(const uint2* __restrict__ cand, const unsigned* __restrict__ count, unsigned limit,
            const float* __restrict__ A, const float* __restrict__ B, int dim, int ld, double* __restrict__ out,
            unsigned first = 0) {
  const int lane = threadIdx.x & 63;
  const unsigned n_cand = min(*count, limit);   // the grid is sized for `limit`; the filter's count lives on the device
  // grid-stride: a launch with fewer workgroups than candidates / 4 walks the list at a lower intensity (the
  // speculative path runs this kernel beside the verification and has half a millisecond of slack for it)
  for (unsigned c = first + blockIdx.x * 4 + (threadIdx.x >> 6); c < n_cand; c += gridDim.x * 4) {
    const uint2 rc = cand[c];
    const double s = nn_exact_sq_dist(A + (size_t)rc.x * ld, B + (size_t)rc.y * ld, ld, lane);
    if (lane == 0) out[c] = sqrt(s);
  }
}

// ---- per-row minima of the candidate list, on the device (the row-sharded NN stage of SURVEY.md section 8(e): no
// candidate list travels to the host).  Same rule as the host loop of nn_run_filter: smallest exact float64 distance,
// ties to the lowest column, ignored pairs skipped, rows without a candidate = (+inf, 0).  Non-negative doubles order
// like their bit patterns, so the minimum is an atomicMin on 64-bit integers; the column is settled in a second pass
// over the candidates that hold their row's minimum.
__global__ void k_nn_rowmin_init(unsigned long long* __restrict__ mn, int* __restrict__ arg, int n, int* status,
                                 unsigned long long* __restrict__ arg64 = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { mn[i] = 0x7FF0000000000000ull; arg[i] = 0x7FFFFFFF; if (arg64) arg64[i] = ~0ull; }
  if (i == 0) *status = 0;
}

__device__ __forceinline__ bool nn_pair_ignored(const int* __restrict__ ign_ptr, const int* __restrict__ ign_col, int r, int col) {
  for (int e = ign_ptr[r]; e < ign_ptr[r + 1]; ++e)
    if (ign_col[e] == col) return true;
  return false;
}

// arg64 (optional, PASS 1): (column << 32 | candidate index) of each row's minimum -- the lowest column among the
// candidates that hold the minimum, and WHICH entry of the list it is (the speculative step maps a match to the
// verification slot of its candidate through it).  A (row, column) pair occurs at most once in the list.
template <int PASS>
__global__ void __launch_bounds__(256)
k_nn_rowmin(const uint2* __restrict__ cand, const unsigned* __restrict__ count, unsigned limit,
            const double* __restrict__ cdist, const int* __restrict__ ign_ptr, const int* __restrict__ ign_col,
            int n_l, int n_r, unsigned long long* __restrict__ mn, int* __restrict__ arg, int* status,
            unsigned long long* __restrict__ arg64 = nullptr) {
  unsigned n = *count;
  if (n > limit) {              // denser than the refinement was sized for: the caller falls back (status 1)
    if (PASS == 0 && blockIdx.x == 0 && threadIdx.x == 0) *status = 1;
    return;
  }
  for (unsigned c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    const uint2 rc = cand[c];
    if ((int)rc.x >= n_l || (int)rc.y >= n_r) continue;
    if (nn_pair_ignored(ign_ptr, ign_col, (int)rc.x, (int)rc.y)) continue;
    const unsigned long long key = (unsigned long long)__double_as_longlong(cdist[c]);
    if (PASS == 0) atomicMin(&mn[rc.x], key);
    else if (key == mn[rc.x]) {
      if (arg64) atomicMin(&arg64[rc.x], ((unsigned long long)rc.y << 32) | (unsigned long long)c);
      else atomicMin(&arg[rc.x], (int)rc.y);
    }
  }
}

__global__ void k_nn_rowmin_finish(int* __restrict__ arg, int n, const unsigned long long* __restrict__ arg64 = nullptr,
                                   int* __restrict__ row_cand = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (arg64) {
    const unsigned long long v = arg64[i];
    arg[i] = v == ~0ull ? 0 : (int)(v >> 32);
    if (row_cand) row_cand[i] = v == ~0ull ? -1 : (int)(unsigned)(v & 0xFFFFFFFFull);
  } else if (arg[i] == 0x7FFFFFFF) {
    arg[i] = 0;
  }
}

// nb_eff[j] = |b_j|^2, +inf for masked / padding columns; *nb_max_bits = the largest finite one (bit pattern; the
// error band of k_nn_select).  The caller zeroes *nb_max_bits.
__global__ void k_nn_fill_norms(float* nb_eff, const float* nb, const uint8_t* mask_other, int n_r, int n_r_pad,
                                unsigned* nb_max_bits) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  if (j < n_r_pad) {
    const bool live = j < n_r && !mask_other[j];
    nb_eff[j] = live ? nb[j] : __int_as_float(0x7F800000);
    if (live && nb[j] < __int_as_float(0x7F800000)) v = nb[j];     // (false for NaN)
  }
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  if ((threadIdx.x & 63) == 0 && v > 0.f) atomicMax(nb_max_bits, __float_as_uint(v));
}


// ---- data_handler.py:191-205 on the device: argsort of the row minima + the sequential walk --------------------------
// Same rule as sf_nn_walk_host below, in three launches and without the host: (1) every 2048-row tile of the minima is
// sorted in LDS on the composite key (float64 bit pattern, row) -- rows at or over the threshold carry the sentinel key
// and sort behind the tile's live rows; (2) a live row's position in the global order = its position in its tile + the
// number of smaller keys in every other tile (a binary search per tile; keys are unique, so the order is the stable
// sort's: ties keep the lowest row first), and the first min(N_l, max_matches_nb) positions claim their column with an
// atomicMin of the position -- the walk's "idx_other already taken" rule (:199-200) keeps exactly the FIRST position of
// each column; (3) one workgroup emits the kept positions in order (an ordered compaction), up to `cap`.
constexpr int WALK_TILE = 2048;
constexpr unsigned long long WALK_SENTINEL = ~0ull;

__device__ __forceinline__ bool walk_less(unsigned long long ka, int ra, unsigned long long kb, int rb) {
  return ka < kb || (ka == kb && ra < rb);
}

__global__ void __launch_bounds__(1024)
k_walk_tile_sort(const double* __restrict__ rm, int n_l, double thr, unsigned long long* __restrict__ keys,
                 int* __restrict__ rows, int* __restrict__ tile_cnt, int* __restrict__ minpos, int n_r) {
  __shared__ unsigned long long sk[WALK_TILE];
  __shared__ int sr[WALK_TILE];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, base = blockIdx.x * WALK_TILE;
  for (int j = blockIdx.x * 1024 + tid; j < n_r; j += gridDim.x * 1024) minpos[j] = 0x7FFFFFFF;
  if (tid == 0) s_cnt = 0;
  for (int e = tid; e < WALK_TILE; e += 1024) {
    const int i = base + e;
    unsigned long long k = WALK_SENTINEL;
    if (i < n_l) {
      const double v = rm[i];
      if (v < thr) k = (unsigned long long)__double_as_longlong(v == 0.0 ? 0.0 : v);      // (-0.0 -> +0.0)
    }
    sk[e] = k;
    sr[e] = i;
  }
  __syncthreads();
  for (int k = 2; k <= WALK_TILE; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int i = 2 * tid - (tid & (j - 1)), l = i + j;       // i has bit j clear
      const bool up = (i & k) == 0;
      const unsigned long long ka = sk[i], kb = sk[l];
      const int ra = sr[i], rb = sr[l];
      if (walk_less(kb, rb, ka, ra) == up) { sk[i] = kb; sk[l] = ka; sr[i] = rb; sr[l] = ra; }
      __syncthreads();
    }
  }
  for (int e = tid; e < WALK_TILE; e += 1024) {
    keys[base + e] = sk[e];
    rows[base + e] = sr[e];
    if (sk[e] != WALK_SENTINEL && (e + 1 == WALK_TILE || sk[e + 1] == WALK_SENTINEL)) s_cnt = e + 1;
  }
  __syncthreads();
  if (tid == 0) tile_cnt[blockIdx.x] = s_cnt;
}

__global__ void __launch_bounds__(256)
k_walk_rank(const unsigned long long* __restrict__ keys, const int* __restrict__ rows, const int* __restrict__ tile_cnt,
            int n_tiles, int lim, const int* __restrict__ row_arg, int n_r, int* __restrict__ sorted_rows,
            int* __restrict__ minpos, int* __restrict__ nu_out) {
  __shared__ int s_part[4];
  const int tid = threadIdx.x;
  const int t = blockIdx.x / (WALK_TILE / 256), p = (blockIdx.x % (WALK_TILE / 256)) * 256 + tid;
  int part = 0;
  for (int j = tid; j < n_tiles; j += 256) part += tile_cnt[j];
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  if ((tid & 63) == 0) s_part[tid >> 6] = part;
  __syncthreads();
  const int nu = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  if (blockIdx.x == 0 && tid == 0) *nu_out = nu;
  if (p >= tile_cnt[t]) return;
  const unsigned long long key = keys[(size_t)t * WALK_TILE + p];
  const int row = rows[(size_t)t * WALK_TILE + p];
  int s = p;
  for (int u = 0; u < n_tiles; ++u) {
    if (u == t) continue;
    const unsigned long long* ku = keys + (size_t)u * WALK_TILE;
    const int* ru = rows + (size_t)u * WALK_TILE;
    int lo = 0, hi = tile_cnt[u];                 // number of entries of tile u below (key, row)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (walk_less(ku[mid], ru[mid], key, row)) lo = mid + 1; else hi = mid;
    }
    s += lo;
  }
  if (s < min(lim, nu)) {
    sorted_rows[s] = row;
    const int io = row_arg[row];
    if ((unsigned)io < (unsigned)n_r) atomicMin(&minpos[io], s);
  }
}

// out_rc / count_block: the match list in the candidate list's layout ((row, column) pairs behind a 64-byte counter
// block: word 0 = matches, word 4 = the accepted-result stream's slot counter, zeroed here) -- what the verification
// kernels take their pairs from; out_matches / out_n / out_status: the same for the caller (host-pinned or device).
__global__ void __launch_bounds__(1024)
k_walk_emit(const int* __restrict__ sorted_rows, const int* __restrict__ nu_p, int lim, const double* __restrict__ rm,
            const int* __restrict__ row_arg, int n_r, const int* __restrict__ minpos, const int* __restrict__ status,
            int cap, uint2* __restrict__ out_rc, unsigned* __restrict__ count_block, sf_match* __restrict__ out_matches,
            int32_t* __restrict__ out_n, int32_t* __restrict__ out_status, const int* __restrict__ row_cand,
            int32_t* __restrict__ out_slot, const unsigned* __restrict__ cand_count, unsigned cand_grid) {
  __shared__ int s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int st = status ? *status : 0;
  // (speculative step: more candidates than verification slots were launched for -- status 2, the caller falls back)
  if (st == 0 && cand_count && *cand_count > cand_grid) st = 2;
  const int S = st ? 0 : min(lim, *nu_p);
  int base = 0;
  for (int s0 = 0; s0 < S && base < cap; s0 += 1024) {
    const int s = s0 + tid;
    bool ok = false;
    int il = 0, io = 0;
    if (s < S) {
      il = sorted_rows[s];
      io = row_arg[il];
      ok = (unsigned)io < (unsigned)n_r && minpos[io] == s;
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { woff += (w < wave) ? s_w[w] : 0; tot += s_w[w]; }
    const int j = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && j < cap) {
      if (out_rc) out_rc[j] = make_uint2((unsigned)il, (unsigned)io);
      if (out_matches) { sf_match m; m.idx_local = il; m.idx_other = io; m.distance = rm[il]; out_matches[j] = m; }
      if (out_slot) out_slot[j] = row_cand ? row_cand[il] : j;
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) {
    const int n = min(base, cap);
    if (count_block) { count_block[0] = (unsigned)n; count_block[4] = 0u; }       // (null for a speculative step: the
                                                                              //  verification runs on the candidate list)
    if (out_n) *out_n = n;
    if (out_status) *out_status = st;
  }
}

}  // namespace

static int nn_reserve(sf_context* c, NNDb& db, int n_total, int ld) {
  const int cap = std::max(128, (n_total + 127) & ~127);
  if (ld != db.ld) {
    // the row pitch changed (sf_nn_reset, then a database of another dimension): the old buffers were sized and
    // zero-padded for the old pitch.  Only an EMPTY database can change its pitch.
    if (db.n != 0) return sf_fail(c, SF_EINVAL, "NetVLAD row pitch changed on a non-empty database");
    SF_HIP(c, hipStreamSynchronize(c->stream));
    if (db.rows.p) (void)hipFree(db.rows.p);
    if (db.norms.p) (void)hipFree(db.norms.p);
    db.rows = Buf(); db.norms = Buf();
    db.cap = 0;
    db.ld = ld;
    db.h_n = -1;
  }
  if (cap <= db.cap) return SF_OK;
  int newcap = std::max(cap, ((db.cap * 2) + 127) & ~127);
  const size_t old_rows = (size_t)db.cap * ld * 4, old_norms = (size_t)db.cap * 4;
  int rc;
  if ((rc = sf_buf_reserve(c, db.rows, (size_t)newcap * ld * 4, true)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, db.norms, (size_t)newcap * 4, true)) != SF_OK) return rc;
  // zero the new tail so tile loads of padding rows read zeros
  SF_HIP(c, hipMemsetAsync((char*)db.rows.p + old_rows, 0, (size_t)newcap * ld * 4 - old_rows, c->stream));
  SF_HIP(c, hipMemsetAsync((char*)db.norms.p + old_norms, 0, (size_t)newcap * 4 - old_norms, c->stream));
  db.cap = newcap;
  return SF_OK;
}

// staging of the per-tick host append: pinned host + device bounce buffers owned by the handle (grow-only).  The
// caller's rows are copied into the pinned block before the call returns (the pointer is only borrowed), the H2D
// copy and the cast kernel are asynchronous; the next append waits for `nn_stage_done` before it overwrites the
// block -- by then (one tick later, find_separators.py:17) the copy has long finished.
static int nn_stage_reserve(sf_context* c, size_t bytes) {
  if (c->nn_stage_busy) {
    SF_HIP(c, hipEventSynchronize(c->nn_stage_done));
    c->nn_stage_busy = false;
  }
  if (!c->nn_stage_done) SF_HIP(c, hipEventCreateWithFlags(&c->nn_stage_done, hipEventDisableTiming));
  if (bytes > c->nn_stage_pinned_bytes) {
    if (c->nn_stage_pinned) (void)hipHostFree(c->nn_stage_pinned);
    c->nn_stage_pinned = nullptr;
    c->nn_stage_pinned_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 16);
    if (hipHostMalloc(&c->nn_stage_pinned, want, hipHostMallocDefault) != hipSuccess)
      return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) failed", want);
    c->nn_stage_pinned_bytes = want;
  }
  return sf_buf_reserve(c, c->nn_stage_dev, std::max<size_t>(bytes, (size_t)1 << 16));
}

// src_kind 0: host float64 rows ; 1: device float32 rows
int sf_nn_append(sf_context* c, NNDb& db, const void* src, int n, int dim, int src_kind) {
  (void)sf_lanes_touch(c, false);     // rows are written through the handle's stream: the second step lane waits for them
  if (n < 0 || dim <= 0) return sf_fail(c, SF_EINVAL, "bad descriptor block %d x %d", n, dim);
  if (n == 0) return SF_OK;
  if (!src) return sf_fail(c, SF_EINVAL, "null descriptor pointer");
  if (c->nn_dim == 0) c->nn_dim = dim;
  if (dim != c->nn_dim)
    return sf_fail(c, SF_EINVAL, "descriptor dimension %d differs from the database's %d (data_handler.py:300-301 reshape)", dim, c->nn_dim);
  SF_HIP(c, hipSetDevice(c->device));
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  int rc = nn_reserve(c, db, db.n + n, ld);
  if (rc != SF_OK) return rc;
  float* dst = (float*)db.rows.p + (size_t)db.n * ld;
  float* nrm = (float*)db.norms.p + db.n;
  if (src_kind == 0) {
    const size_t bytes = (size_t)n * dim * 8;
    if ((rc = nn_stage_reserve(c, bytes)) != SF_OK) return rc;
    memcpy(c->nn_stage_pinned, src, bytes);
    SF_HIP(c, hipMemcpyAsync(c->nn_stage_dev.p, c->nn_stage_pinned, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_nn_cast_rows, dim3((n + 3) / 4), dim3(256), 0, c->stream, (const double*)c->nn_stage_dev.p, dst,
                       nrm, n, dim, ld);
    SF_HIP(c, hipGetLastError());
    SF_HIP(c, hipEventRecord(c->nn_stage_done, c->stream));
    c->nn_stage_busy = true;
  } else if (src_kind == 2) {
    hipLaunchKernelGGL(k_nn_copy_rows_f16, dim3((n + 3) / 4), dim3(256), 0, c->stream, (const _Float16*)src, dst, nrm, n, dim, ld);
    SF_HIP(c, hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_nn_copy_rows, dim3((n + 3) / 4), dim3(256), 0, c->stream, (const float*)src, dst, nrm, n, dim, ld);
    SF_HIP(c, hipGetLastError());
  }
  db.n += n;
  c->masks_dirty = true;
  return SF_OK;
}


// ---- fp16 filter path -----------------------------------------------------------------------------
static int nn_prepare_f16(sf_context* c, NNDb& db, int ld, int ld16, int kprefix) {
  if (db.h_n == db.n && db.h_ld == ld16 && db.h_kprefix == kprefix) return SF_OK;
  int rc;
  c->prep_count += 1;          // (state every step lane reads is being rebuilt: sf_step_issue orders the lanes behind it)
  if ((rc = sf_buf_reserve(c, c->nn_scalar, 64)) != SF_OK) return rc;
  SF_HIP(c, hipMemsetAsync(c->nn_scalar.p, 0, 64, c->stream));
  hipLaunchKernelGGL(k_nn_maxabs, dim3(1024), dim3(256), 0, c->stream, (const float*)db.rows.p, (size_t)db.n * ld,
                     (unsigned*)c->nn_scalar.p);
  unsigned bits = 0;
  SF_HIP(c, hipMemcpyAsync(&bits, c->nn_scalar.p, 4, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  float maxabs;
  memcpy(&maxabs, &bits, 4);
  float scale = 1.f;
  if (maxabs > 0.f && std::isfinite(maxabs)) {
    int e;
    frexpf(32768.f / maxabs, &e);          // 32768/maxabs = m * 2^e, m in [0.5, 1)
    e = std::max(-60, std::min(60, e - 1));
    scale = ldexpf(1.f, e);                // largest power of two with maxabs * scale <= 32768
  }
  const int n_pad = (db.n + 127) & ~127;
  const size_t bytes = (size_t)n_pad * ld16 * 2;
  if ((rc = sf_buf_reserve(c, db.rows_h, bytes)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, db.norms_k, (size_t)n_pad * 4)) != SF_OK) return rc;
  SF_HIP(c, hipMemsetAsync(db.rows_h.p, 0, bytes, c->stream));
  SF_HIP(c, hipMemsetAsync(db.norms_k.p, 0, (size_t)n_pad * 4, c->stream));
  hipLaunchKernelGGL(k_nn_to_f16, dim3((db.n + 3) / 4), dim3(256), 0, c->stream, (const float*)db.rows.p,
                     (_Float16*)db.rows_h.p, (float*)db.norms_k.p, db.n, ld, ld16, kprefix, scale);
  SF_HIP(c, hipGetLastError());
  db.h_n = db.n;
  db.h_ld = ld16;
  db.h_kprefix = kprefix;
  db.h_scale = scale;
  return SF_OK;
}

// Returns SF_OK with *done = 1 when the filter path produced the row minima; *done = 0 means the
// candidate buffer overflowed (threshold too loose for a sparse filter) -> caller runs the exact path.
// SF_NN_TRACE=1: host-side stage times of one query on stderr (diagnostic)
struct NnTrace {
  bool on;
  std::chrono::steady_clock::time_point t;
  NnTrace() : on(getenv("SF_NN_TRACE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* what, long long n = -1) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[sf nn] %-22s %8.3f ms", what, std::chrono::duration<double, std::milli>(now - t).count());
    if (n >= 0) fprintf(stderr, "  (%lld)", n);
    fputc('\n', stderr);
    t = now;
  }
};

static int nn_pinned_reserve(sf_context* c, size_t need) {
  if (need <= c->nn_pinned_bytes) return SF_OK;
  if (c->nn_pinned) (void)hipHostFree(c->nn_pinned);
  c->nn_pinned = nullptr;
  c->nn_pinned_bytes = 0;
  const size_t want = need + need / 2;
  if (hipHostMalloc(&c->nn_pinned, want, hipHostMallocDefault) != hipSuccess)
    return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) failed", want);
  c->nn_pinned_bytes = want;
  return SF_OK;
}

struct NnFilterBufs {
  unsigned cap = 0;
  float2* rowc = nullptr;
  float2* colc = nullptr;
  unsigned* count = nullptr;     // the counter block of the LAST launch (two 64-byte blocks alternate)
  uint2* cand = nullptr;
  double* cdist = nullptr;
};

static int nn_filter_reserve(sf_context* c, NnFilterBufs& fb) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n;
  const int n_l_pad = (n_l + NN_BM - 1) / NN_BM * NN_BM, n_r_pad = (n_r + NN_BN - 1) / NN_BN * NN_BN;
  int rc;
  const unsigned cap = (unsigned)std::max<size_t>((size_t)1 << 20, (size_t)64 * n_l);
  {
    const void* before = c->nn_cand.p;
    if ((rc = sf_buf_reserve(c, c->nn_cand, (size_t)cap * 16 + 128)) != SF_OK) return rc;
    if (c->nn_cand.p != before) c->nn_count_primed = false;      // (a fresh buffer: nobody zeroed its counter blocks)
  }
  if ((rc = sf_buf_reserve(c, c->nn_rowmin, (size_t)(n_l_pad + n_r_pad) * 8)) != SF_OK) return rc;
  fb.rowc = (float2*)c->nn_rowmin.p;
  fb.colc = fb.rowc + n_l_pad;
  fb.count = (unsigned*)c->nn_cand.p;
  fb.cand = (uint2*)((char*)c->nn_cand.p + 128);
  fb.cdist = (double*)((char*)c->nn_cand.p + 128 + (size_t)cap * 8);
  fb.cap = cap;

  return SF_OK;
}

// One launch of the stage-1 filter at prefix length `kdims` (ladder level `level`): fp16 copies, coefficients, kernel.
// Leaves the candidate list in fb.cand and its length in *fb.count (device).
static int nn_filter_launch(sf_context* c, NnFilterBufs& fb, int level, int kdims, NnTrace* trp) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  const int n_l_pad = (n_l + NN_BM - 1) / NN_BM * NN_BM, n_r_pad = (n_r + NN_BN - 1) / NN_BN * NN_BN;
  const unsigned cap = fb.cap;
  float2* const rowc = fb.rowc;
  float2* const colc = fb.colc;
  uint2* const cand = fb.cand;
  unsigned* count;
  NnTrace& tr = *trp;
  const double thr = c->params.netvlad_distance;
  float thr2 = (float)(thr * thr);
  thr2 = nextafterf(thr2, INFINITY);
  int rc;
    // the fp16 copy of a prefix level is stored COMPACT (pitch = prefix length): a 128-row operand tile is
    // then one contiguous 32 / 128 KB block instead of 128 pieces 2 * dim bytes apart
    const int pitch16 = kdims;
    if ((rc = nn_prepare_f16(c, c->nn_local, ld, pitch16, kdims)) != SF_OK) return rc;
    if ((rc = nn_prepare_f16(c, c->nn_recv, ld, pitch16, kdims)) != SF_OK) return rc;
    if (tr.on) { (void)hipStreamSynchronize(c->stream); tr.mark("prepare f16", kdims); }
    // |dot16 - dot32| <= (2^-10 (1 + 2^-11) + k 2^-24) * ||a|| ||b||  (operand rounding + fp32 accumulation)
    const float eps_rel = (float)(ldexp(1.0, -10) * 1.001 + (double)kdims * ldexp(1.0, -24));
    const float scale = c->nn_local.h_scale * c->nn_recv.h_scale;   // product of two powers of two: exact
    count = (unsigned*)((char*)c->nn_cand.p + 64 * c->nn_count_idx);
    unsigned* const count_next = (unsigned*)((char*)c->nn_cand.p + 64 * (c->nn_count_idx ^ 1));
    if (!c->nn_count_primed) SF_HIP(c, hipMemsetAsync(count, 0, 64, c->stream));
    c->nn_count_idx ^= 1;
    c->nn_count_primed = false;
    // the per-row / per-column coefficients depend only on the norms, the masks, the threshold and the
    // prefix level: rebuilt when one of them changed, not per query
    const bool coef_ok = c->nn_coef_level == level && c->nn_coef_nl == n_l && c->nn_coef_nr == n_r &&
                         c->nn_coef_thr == thr && c->nn_coef_ptr == (const void*)rowc && c->nn_coef_scale == scale;
    const float delta = 2e-6f + (float)((double)(kdims + 8) * ldexp(1.0, -24));
    if (!coef_ok) c->prep_count += 1;
    if (!coef_ok)
    hipLaunchKernelGGL(k_nn_filter_row_coef, dim3((n_l_pad + 255) / 256), dim3(256), 0, c->stream, rowc,
                       (const float*)c->nn_local.norms_k.p, (const uint8_t*)c->d_mask_local.p, n_l, n_l_pad,
                       0.5f * scale, thr2, eps_rel * c->nn_local.h_scale * 64.f, delta);
    if (!coef_ok) {
      hipLaunchKernelGGL(k_nn_filter_col_coef, dim3((n_r_pad + 255) / 256), dim3(256), 0, c->stream, colc,
                         (const float*)c->nn_recv.norms_k.p, (const uint8_t*)c->d_mask_other.p, n_r, n_r_pad,
                         0.5f * scale, c->nn_recv.h_scale / 64.f, delta);
      c->nn_coef_level = level; c->nn_coef_nl = n_l; c->nn_coef_nr = n_r; c->nn_coef_thr = thr;
      c->nn_coef_ptr = (const void*)rowc;
      c->nn_coef_scale = scale;
    }
    sf_prof_begin(c, SF_K_NN_FILTER);
    if (kdims == 128 && getenv("SF_NN_K128_OFF") == nullptr) {
      // resident row panel, strips of column tiles: about two workgroups per CU in one wave of the grid
      const int gx = n_r_pad / NN_BN, gy = n_l_pad / NN_BM;
      if (c->cu_count <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || v <= 0) v = 256;
        c->cu_count = v;
      }
      const int strips = std::max(1, std::min(gx, (2 * c->cu_count) / std::max(1, gy)));
      const int tps = (gx + strips - 1) / strips;
      static const bool k128_lds_panel = getenv("SF_NN_K128_LDS_PANEL") != nullptr;   // (round-2 form, for A/B runs)
#ifdef SF_NN_ABLATION
      // timing-only forms (make NN_ABLATION=1; tools/nn_filter_time.py): bit 0 no hit scan, bit 1 no LDS operand reads,
      // bit 2 no DMA of the next tile, bit 3 no tile loop, bit 4 no prologue DMA -- never compiled into the product build
      static const int k128_abl = getenv("SF_NN_K128_ABL") ? atoi(getenv("SF_NN_K128_ABL")) : 0;
      auto kern = k128_abl == 1 ? k_nn_filter_f16_k128r<1> : k128_abl == 3 ? k_nn_filter_f16_k128r<3>
                  : k128_abl == 5 ? k_nn_filter_f16_k128r<5> : k128_abl == 7 ? k_nn_filter_f16_k128r<7>
                  : k128_abl == 8 ? k_nn_filter_f16_k128r<8> : k128_abl == 24 ? k_nn_filter_f16_k128r<24>
                  : k_nn_filter_f16_k128r<0>;
#else
      auto kern = k_nn_filter_f16_k128r<0>;
#endif
      if (!c->nn_k128_attr) {
        SF_HIP(c, hipFuncSetAttribute((const void*)k_nn_filter_f16_k128, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      NN_K128_LDS));
        c->nn_k128_attr = true;
      }
      if (k128_lds_panel)
        hipLaunchKernelGGL(k_nn_filter_f16_k128, dim3(gy * ((gx + tps - 1) / tps)), dim3(256), NN_K128_LDS, c->stream,
                           (const _Float16*)c->nn_local.rows_h.p, (const _Float16*)c->nn_recv.rows_h.p, rowc, colc, gx,
                           gy, tps, cand, count, cap, count_next);
      else      // (static LDS: two tile buffers + the hit list, 74 KB)
        hipLaunchKernelGGL(kern, dim3(gy * ((gx + tps - 1) / tps)), dim3(256), 0, c->stream,
                           (const _Float16*)c->nn_local.rows_h.p, (const _Float16*)c->nn_recv.rows_h.p, rowc, colc, gx,
                           gy, tps, cand, count, cap, count_next);
      c->nn_count_primed = true;                 // (this launch zeroes the other block for the next one)
    } else if (nn_t256_on() && kdims % 64 == 0 && kdims >= 256) {
      static bool attr256 = false;      // (one device per process in every deployment of this library; set once)
      if (!attr256) {
        SF_HIP(c, hipFuncSetAttribute((const void*)k_nn_filter_f16_t256, hipFuncAttributeMaxDynamicSharedMemorySize, NN256_LDS));
        attr256 = true;
      }
      const int gx256 = (n_r_pad + 255) / 256, gy256 = (n_l_pad + 255) / 256;
      hipLaunchKernelGGL(k_nn_filter_f16_t256, dim3(gx256 * gy256), dim3(512), NN256_LDS, c->stream,
                         (const _Float16*)c->nn_local.rows_h.p, (const _Float16*)c->nn_recv.rows_h.p, rowc, colc,
                         pitch16, kdims, gx256, gy256, n_l_pad, n_r_pad, cand, count, cap);
    } else {
      hipLaunchKernelGGL(k_nn_filter_f16, dim3((n_r_pad / NN_BN) * (n_l_pad / NN_BM)), dim3(256), 0, c->stream,
                         (const _Float16*)c->nn_local.rows_h.p, (const _Float16*)c->nn_recv.rows_h.p, rowc, colc,
                         pitch16, kdims, n_r_pad / NN_BN, n_l_pad / NN_BM, cand, count, cap);
    }
    sf_prof_end(c, SF_K_NN_FILTER);
    SF_HIP(c, hipGetLastError());
    fb.count = count;
  return SF_OK;
}

static int nn_run_filter(sf_context* c, int* done) {
  *done = 0;
  NnTrace tr;
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  const int ld16 = (dim + 63) / 64 * 64;
  int rc;
  // Stage 1 contracts only a PREFIX of the descriptor: the squared distance over the first k
  // dimensions is a lower bound of the full squared distance, so "prefix distance (within the fp16
  // error band) under the threshold" is a necessary condition.  PCA-whitened NetVLAD spreads its
  // energy evenly (the reference itself ranks on a 128-dim prefix, data_handler.py:157-158), so an
  // eighth of the dimensions already rejects everything but real neighbours.  If the candidate
  // buffer overflows the full length is tried, and after that the exact fp32-ranking path.
  const int kfull = ld16;
  // prefix ladder: 128 and 512 dimensions when they are at most a quarter of the descriptor, then the
  // full length.  The handle remembers the level that last produced a sparse candidate set.
  int levels[3], n_levels = 0;
  if (128 * 4 <= dim) levels[n_levels++] = 128;
  if (512 * 4 <= dim) levels[n_levels++] = 512;
  levels[n_levels++] = kfull;
  int level = std::min(std::max(c->nn_level, 0), n_levels - 1);
  if (c->nn_force_full) {
    level = n_levels - 1;      // SF_OPT_NN_FULL_FILTER: no prefix level (the worst case of the adaptive ladder)
  } else if (level > 0 && c->nn_level_cooldown == 0) {   // now and then re-try the cheaper level
    --level;
    c->nn_level_cooldown = 32;
  } else if (c->nn_level_cooldown > 0) {
    --c->nn_level_cooldown;
  }
  NnFilterBufs fb;
  if ((rc = nn_filter_reserve(c, fb)) != SF_OK) return rc;
  const unsigned cap = fb.cap;
  uint2* const cand = fb.cand;
  double* const cdist = fb.cdist;
  unsigned* count = fb.count;
  unsigned n_cand = 0;
  bool ok = false;
  uint2* h_cand = nullptr;     // pinned staging of the candidate list (pageable D2H copies are staged and slow)
  double* h_dist = nullptr;
  for (; level < n_levels && !ok; ++level) {
    const int kdims = levels[level];
    if ((rc = nn_filter_launch(c, fb, level, kdims, &tr)) != SF_OK) return rc;
    count = fb.count;
    const bool prefix_level = level < n_levels - 1;
    // a dense prefix result would make the exact refinement the expensive part: insist on a sparse
    // candidate set from a prefix level, accept anything that fits the buffer from the full-length level
    const unsigned limit = prefix_level ? (unsigned)(8 * (size_t)n_l + 4096) : cap;
    if (prefix_level) {
      // no host round trip between the filter and the refinement: the refine grid is sized for the
      // sparse limit and reads the count on the device; count, candidates and distances come back
      // behind ONE synchronisation (a speculative prefix of 2N + 1024 entries, the rest if needed)
      if ((rc = nn_pinned_reserve(c, (size_t)limit * 16 + 64)) != SF_OK) return rc;
      h_cand = (uint2*)((char*)c->nn_pinned + 64);
      h_dist = (double*)((char*)c->nn_pinned + 64 + (size_t)limit * 8);
      const unsigned spec = std::min<unsigned>(limit, (unsigned)(2 * (size_t)n_l + 1024));
      // With a speculative verification requested (sf_find_matches_and_verify_device) the handle's stream goes
      // from the filter straight into the verification of every candidate -- which needs the candidates' (row,
      // column), not their exact distances -- while the exact re-evaluation (HBM-bound) and the copies back to
      // the host run on a second stream beside it; the host's row minima / sort / walk then run beside the
      // verification kernels too.
      const bool speculate = c->spec.requested && !c->spec.launched;
      hipStream_t cs = c->stream;
      if (speculate) {
        SF_HIP(c, hipEventRecord(c->spec.ev_refined, c->stream));          // = the filter has finished
        SF_HIP(c, hipStreamWaitEvent(c->spec.copy_stream, c->spec.ev_refined, 0));
        cs = c->spec.copy_stream;
        if ((rc = sf_spec_launch(c, cand, count)) != SF_OK) return rc;     // handle's stream: pair list + verification
        c->spec.launched = true;
      }
      {
        const hipStream_t main_stream = c->stream;
        c->stream = cs;                      // (sf_prof_begin / _end record on the stream the kernel runs on)
        sf_prof_begin(c, SF_K_NN_REFINE);
        // sized like the speculative copy below (2N + 1024 candidates): a grid for the whole sparse limit
        // (8N + 4096) is 4/5 empty workgroups that the dispatcher still has to walk through -- beside the
        // verification kernel in the speculative path; the rare tail is re-evaluated once its size is known
        // beside a speculative verification of full-size frames the re-evaluation has ~10x its own run time of
        // slack: one workgroup per CU walks the list (grid-stride) instead of a workgroup per four candidates, which
        // leaves the HBM and the dispatcher to the verification kernel's first third (k_verify_fused 0.60 -> 0.58 ms,
        // the re-evaluation 0.09 -> 0.16 ms in its shadow)
        unsigned refine_wgs = (spec + 3) / 4;
        if (speculate && c->store.kcap >= 256) {
          if (c->n_cus <= 0) {
            int v = 0;
            c->n_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) ? v : 256;
          }
          refine_wgs = std::min(refine_wgs, (unsigned)c->n_cus);
        }
        hipLaunchKernelGGL(k_nn_refine, dim3(refine_wgs), dim3(256), 0, cs, cand, count, spec,
                           (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld, cdist, 0u);
        sf_prof_end(c, SF_K_NN_REFINE);
        c->stream = main_stream;
      }
      SF_HIP(c, hipMemcpyAsync(c->nn_pinned, count, 4, hipMemcpyDeviceToHost, cs));
      SF_HIP(c, hipMemcpyAsync(h_cand, cand, (size_t)spec * 8, hipMemcpyDeviceToHost, cs));
      SF_HIP(c, hipMemcpyAsync(h_dist, cdist, (size_t)spec * 8, hipMemcpyDeviceToHost, cs));
      if (speculate) {
        SF_HIP(c, hipEventRecord(c->spec.ev_copied, cs));
        SF_HIP(c, hipEventSynchronize(c->spec.ev_copied));
      } else {
        SF_HIP(c, hipStreamSynchronize(c->stream));
      }
      n_cand = *(const unsigned*)c->nn_pinned;
      tr.mark("filter + refine + D2H", n_cand);
      ok = n_cand <= limit;
      if (speculate) c->spec.valid = ok && n_cand <= c->spec.grid;
      if (ok && n_cand > spec) {
        // (rare) the tail of the candidate list; on the copy stream when the handle's stream is already busy
        // with the speculative verification
        hipLaunchKernelGGL(k_nn_refine, dim3((n_cand - spec + 3) / 4), dim3(256), 0, cs, cand, count, n_cand,
                           (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld, cdist, spec);
        SF_HIP(c, hipMemcpyAsync(h_cand + spec, cand + spec, (size_t)(n_cand - spec) * 8, hipMemcpyDeviceToHost, cs));
        SF_HIP(c, hipMemcpyAsync(h_dist + spec, cdist + spec, (size_t)(n_cand - spec) * 8, hipMemcpyDeviceToHost, cs));
        SF_HIP(c, hipStreamSynchronize(cs));
      }
    } else {
      SF_HIP(c, hipMemcpyAsync(&n_cand, count, 4, hipMemcpyDeviceToHost, c->stream));
      SF_HIP(c, hipStreamSynchronize(c->stream));
      tr.mark("filter + count", n_cand);
      ok = n_cand <= limit;
      if (ok) {
        if ((rc = nn_pinned_reserve(c, (size_t)n_cand * 16 + 64)) != SF_OK) return rc;
        h_cand = (uint2*)((char*)c->nn_pinned + 64);
        h_dist = (double*)((char*)c->nn_pinned + 64 + (size_t)n_cand * 8);
        if (n_cand) {
          sf_prof_begin(c, SF_K_NN_REFINE);
          hipLaunchKernelGGL(k_nn_refine, dim3((n_cand + 3) / 4), dim3(256), 0, c->stream, cand, count, n_cand,
                             (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld, cdist);
          sf_prof_end(c, SF_K_NN_REFINE);
          SF_HIP(c, hipMemcpyAsync(h_cand, cand, (size_t)n_cand * 8, hipMemcpyDeviceToHost, c->stream));
          SF_HIP(c, hipMemcpyAsync(h_dist, cdist, (size_t)n_cand * 8, hipMemcpyDeviceToHost, c->stream));
          SF_HIP(c, hipStreamSynchronize(c->stream));
        }
      }
    }
    if (ok) {
      if (!c->nn_force_full) c->nn_level = level;
      c->nn_last_kdims = kdims;
    }
  }
  if (!ok) {   // too dense for the filter: exact path
    c->nn_last_kdims = 0;
    return SF_OK;
  }
  tr.mark("refine + D2H");
  // per-row minimum over the exact candidate distances (ties: lowest column), ignored pairs skipped.
  // Rows without a candidate have their true minimum >= netvlad_distance: reported as +inf.
  c->last_row_min.assign(n_l, (double)INFINITY);
  c->last_row_arg.assign(n_l, 0);
  c->last_row_cand.assign(n_l, -1);
  std::vector<std::pair<int, int>> ign;
  for (size_t e = 0; e + 1 < c->ignored.size(); e += 2) ign.push_back({c->ignored[e], c->ignored[e + 1]});
  std::sort(ign.begin(), ign.end());
  for (unsigned i = 0; i < n_cand; ++i) {
    const int r = (int)h_cand[i].x, col = (int)h_cand[i].y;
    if (r >= n_l || col >= n_r) continue;
    if (!ign.empty() && std::binary_search(ign.begin(), ign.end(), std::make_pair(r, col))) continue;
    const double d = h_dist[i];
    if (d < c->last_row_min[r] || (d == c->last_row_min[r] && col < c->last_row_arg[r])) {
      c->last_row_min[r] = d;
      c->last_row_arg[r] = col;
      c->last_row_cand[r] = (int32_t)i;
    }
  }
  tr.mark("host row minima");
  // a candidate admitted only by the error band may still be >= the threshold: that is fine, the
  // walk compares the exact float64 value
  *done = 1;
  return SF_OK;
}

// data_handler.py:191-205 on explicit per-row minima: argsort of the row minima, then the sequential walk.  Host
// only (no GPU work): sf_nn_run ends with it, and sf_nn_walk exposes it for the row-sharded NN stage of a multi-GPU
// node, where every rank runs the identical walk on the all-gathered minima.
int sf_nn_walk_host(sf_context* c, const double* rm, const int32_t* row_arg, int n_l, int n_r, double thr_d,
                    int max_matches_nb, sf_match* out, int cap, int* n_out) {
  // data_handler.py:191-205: argsort of the row minima, then the sequential walk.
  // Only rows whose minimum is under the threshold can be accepted, and they sort in front of all
  // others; once they are exhausted the reference's loop can only `continue` or `break`.  So the
  // sort is restricted to those rows: LSD radix sort on the float64 bit patterns (non-negative
  // doubles order like unsigned integers; the sort is stable, so ties keep the lowest row first).
  NnTrace tr2;
  std::vector<uint64_t>& keys = c->nn_sort_keys;
  std::vector<int32_t>& rows = c->nn_sort_rows;
  keys.clear();
  rows.clear();
  for (int i = 0; i < n_l; ++i) {
    if (rm[i] < thr_d) {
      uint64_t b;
      const double v = rm[i] == 0.0 ? 0.0 : rm[i];   // -0.0 -> +0.0
      memcpy(&b, &v, 8);
      keys.push_back(b);
      rows.push_back(i);
    }
  }
  const size_t nu = keys.size();
  if (nu > 1) {
    // LSD radix sort, 11-bit digits (6 passes cover 64 bits); all digit histograms come from ONE read
    // of the keys, and a pass whose digit is identical in every key is skipped
    constexpr int DIG = 11, NB = 1 << DIG, NP = (64 + DIG - 1) / DIG;
    std::vector<uint64_t>& k2 = c->nn_sort_keys2;
    std::vector<int32_t>& r2 = c->nn_sort_rows2;
    k2.resize(nu);
    r2.resize(nu);
    std::vector<uint32_t>& hist = c->nn_sort_hist;
    hist.assign((size_t)NP * NB, 0u);
    for (size_t i = 0; i < nu; ++i) {
      const uint64_t k = keys[i];
#pragma unroll
      for (int p = 0; p < NP; ++p) hist[(size_t)p * NB + ((k >> (p * DIG)) & (NB - 1))]++;
    }
    for (int p = 0; p < NP; ++p) {
      uint32_t* h = hist.data() + (size_t)p * NB;
      const int shift = p * DIG;
      if (h[(keys[0] >> shift) & (NB - 1)] == nu) continue;   // this digit is identical in every key
      uint32_t run = 0;
      for (int b2 = 0; b2 < NB; ++b2) { const uint32_t cnt = h[b2]; h[b2] = run; run += cnt; }
      for (size_t i = 0; i < nu; ++i) {
        const uint32_t pos = h[(keys[i] >> shift) & (NB - 1)]++;
        k2[pos] = keys[i];
        r2[pos] = rows[i];
      }
      keys.swap(k2);
      rows.swap(r2);
    }
  }
  tr2.mark("host sort", (long long)nu);
  const int lim = std::min(n_l, max_matches_nb);
  int n = 0;
  std::vector<uint8_t>& taken = c->nn_taken;
  taken.assign(n_r, 0);
  for (int s = 0; s < lim && s < (int)nu; ++s) {
    const int il = rows[s], io = row_arg[il];
    if (io < 0 || io >= n_r) continue;   // (caller-provided minima: never index outside the column range)
    if (taken[io]) continue;                                  // :199-200 (slot still consumed)
    // rm[il] < netvlad_distance holds for every row kept above (:202-203)
    if (n < cap) { out[n].idx_local = il; out[n].idx_other = io; out[n].distance = rm[il]; }
    taken[io] = 1;
    ++n;
    if (n >= cap) break;
  }
  *n_out = std::min(n, cap);
  tr2.mark("host walk", n);
  return SF_OK;
}

// The same walk on the device (kernels above), asynchronous on the handle's stream: d_row_min / d_row_arg are DEVICE
// arrays (sf_nn_row_minima_dev's outputs, or the all-gathered minima of a row-sharded node), d_status (or null) a device
// word that voids the walk when non-zero (the filter's "too dense" report).  Scratch belongs to the handle (per step lane).
int sf_nn_walk_dev(sf_context* c, const double* d_row_min, const int32_t* d_row_arg, const int32_t* d_status, int n_l,
                   int n_r, double thr_d, int max_matches_nb, int cap, void* d_match_rc, unsigned* d_count_block,
                   sf_match* out_matches, int32_t* out_n, int32_t* out_status, const int32_t* d_row_cand,
                   int32_t* out_slot, const unsigned* d_cand_count, unsigned cand_grid) {
  if (n_l <= 0 || n_r <= 0 || cap < 0) return sf_fail(c, SF_EINVAL, "device walk over %d x %d minima, cap %d", n_l, n_r, cap);
  const int n_tiles = (n_l + WALK_TILE - 1) / WALK_TILE;
  const size_t n_pad = (size_t)n_tiles * WALK_TILE;
  // keys | rows | sorted rows | minpos | tile counts | nu
  const size_t off_rows = n_pad * 8, off_sorted = off_rows + n_pad * 4, off_minpos = off_sorted + n_pad * 4,
               off_cnt = off_minpos + (((size_t)n_r * 4 + 63) & ~(size_t)63), off_nu = off_cnt + (((size_t)n_tiles * 4 + 63) & ~(size_t)63);
  int rc;
  if ((rc = sf_buf_reserve(c, c->walk_scratch, off_nu + 64)) != SF_OK) return rc;
  char* w = (char*)c->walk_scratch.p;
  unsigned long long* keys = (unsigned long long*)w;
  int* rows = (int*)(w + off_rows);
  int* sorted = (int*)(w + off_sorted);
  int* minpos = (int*)(w + off_minpos);
  int* tile_cnt = (int*)(w + off_cnt);
  int* nu = (int*)(w + off_nu);
  const int lim = std::min(n_l, max_matches_nb);
  sf_prof_begin(c, SF_K_NN_WALK);
  hipLaunchKernelGGL(k_walk_tile_sort, dim3(n_tiles), dim3(1024), 0, c->stream, d_row_min, n_l, thr_d, keys, rows, tile_cnt,
                     minpos, n_r);
  hipLaunchKernelGGL(k_walk_rank, dim3(n_tiles * (WALK_TILE / 256)), dim3(256), 0, c->stream, keys, rows, tile_cnt, n_tiles,
                     lim, d_row_arg, n_r, sorted, minpos, nu);
  hipLaunchKernelGGL(k_walk_emit, dim3(1), dim3(1024), 0, c->stream, sorted, nu, lim, d_row_min, d_row_arg, n_r, minpos,
                     d_status, cap, (uint2*)d_match_rc, d_count_block, out_matches, out_n, out_status, d_row_cand, out_slot,
                     d_cand_count, cand_grid);
  sf_prof_end(c, SF_K_NN_WALK);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

// masks / ignore CSR on the device (rebuilt only when they changed)
static int nn_sync_masks(sf_context* c) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n;
  const int n_l_pad = (n_l + NN_BM - 1) / NN_BM * NN_BM, n_r_pad = (n_r + NN_BN - 1) / NN_BN * NN_BN;
  int rc;
  c->mask_local.resize(n_l, 0);
  c->mask_other.resize(n_r, 0);
  if (c->masks_dirty) {
    c->prep_count += 1;
    if ((rc = sf_buf_reserve(c, c->d_mask_local, (size_t)n_l_pad)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->d_mask_other, (size_t)n_r_pad)) != SF_OK) return rc;
    std::vector<int> ptr(n_l_pad + 1, 0), col(std::max<size_t>(1, c->ignored.size() / 2));
    for (size_t e = 0; e + 1 < c->ignored.size(); e += 2) ptr[c->ignored[e] + 1]++;
    for (int i = 0; i < n_l_pad; ++i) ptr[i + 1] += ptr[i];
    {
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (size_t e = 0; e + 1 < c->ignored.size(); e += 2) col[fill[c->ignored[e]]++] = c->ignored[e + 1];
    }
    if ((rc = sf_buf_reserve(c, c->d_ign_ptr, ptr.size() * 4)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->d_ign_col, col.size() * 4)) != SF_OK) return rc;
    SF_HIP(c, hipMemcpyAsync(c->d_mask_local.p, c->mask_local.data(), n_l, hipMemcpyHostToDevice, c->stream));
    SF_HIP(c, hipMemcpyAsync(c->d_mask_other.p, c->mask_other.data(), n_r, hipMemcpyHostToDevice, c->stream));
    SF_HIP(c, hipMemcpyAsync(c->d_ign_ptr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, c->stream));
    SF_HIP(c, hipMemcpyAsync(c->d_ign_col.p, col.data(), col.size() * 4, hipMemcpyHostToDevice, c->stream));
    SF_HIP(c, hipStreamSynchronize(c->stream));  // host vectors go out of scope
    c->masks_dirty = false;
    c->nn_coef_level = -1;     // masks / database changed: the filter coefficients must be rebuilt
  }
  return SF_OK;
}

int sf_nn_run(sf_context* c, sf_match* out, int cap, int* n_out) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  if (c->params.nn_precision != 0 && c->params.nn_precision != 1)
    return sf_fail(c, SF_EINVAL, "nn_precision %d unknown (0 = fp32 exact ranking, 1 = fp16 filter + exact refine)", c->params.nn_precision);
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  const int n_l_pad = (n_l + NN_BM - 1) / NN_BM * NN_BM, n_r_pad = (n_r + NN_BN - 1) / NN_BN * NN_BN;
  const int n_strips = n_r_pad / 64;
  int rc;
  if ((rc = nn_sync_masks(c)) != SF_OK) return rc;
  int filtered = 0;
  c->nn_last_kdims = 0;
  if (c->params.nn_precision == 1) {
    if ((rc = nn_run_filter(c, &filtered)) != SF_OK) return rc;
  }
  if (!filtered) {
  c->spec.valid = false;   // the exact path has no candidate list
  c->last_row_cand.clear();
  c->nn_coef_level = -1;   // the exact path re-uses the coefficient buffer for its partial minima
  // workspace: partial minima, effective column norms, per-row results
  const size_t part_bytes = (size_t)n_strips * n_l_pad * 16;   // best + second-best key per (strip, row)
  if ((rc = sf_buf_reserve(c, c->nn_scalar, 64)) != SF_OK) return rc;
  unsigned* nb_max_bits = (unsigned*)((char*)c->nn_scalar.p + 16);
  SF_HIP(c, hipMemsetAsync(nb_max_bits, 0, 4, c->stream));
  if ((rc = sf_buf_reserve(c, c->nn_rowmin, part_bytes + (size_t)n_r_pad * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->nn_exact, (size_t)n_l * 12)) != SF_OK) return rc;
  unsigned long long* part = (unsigned long long*)c->nn_rowmin.p;
  float* nb_eff = (float*)((char*)c->nn_rowmin.p + part_bytes);
  double* d_dist = (double*)c->nn_exact.p;
  int* d_idx = (int*)((char*)c->nn_exact.p + (size_t)n_l * 8);

  hipLaunchKernelGGL(k_nn_fill_norms, dim3((n_r_pad + 255) / 256), dim3(256), 0, c->stream, nb_eff,
                     (const float*)c->nn_recv.norms.p, (const uint8_t*)c->d_mask_other.p, n_r, n_r_pad, nb_max_bits);
  sf_prof_begin(c, SF_K_NN);
  hipLaunchKernelGGL(k_nn_argmin, dim3((n_r_pad / NN_BN) * (n_l_pad / NN_BM)), dim3(256), 0, c->stream,
                     (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p,
                     (const float*)c->nn_local.norms.p, nb_eff, (const int*)c->d_ign_ptr.p,
                     (const int*)c->d_ign_col.p, part, n_l_pad, ld, n_r_pad / NN_BN, n_l_pad / NN_BM);
  sf_prof_end(c, SF_K_NN);
  sf_prof_begin(c, SF_K_NN_SELECT);
  hipLaunchKernelGGL(k_nn_select, dim3((n_l + 3) / 4), dim3(256), 0, c->stream, part, n_strips, n_l, n_l_pad,
                     (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld,
                     (const uint8_t*)c->d_mask_local.p, (const float*)c->nn_local.norms.p, nb_eff, nb_max_bits,
                     (const int*)c->d_ign_ptr.p, (const int*)c->d_ign_col.p, n_r, d_dist, d_idx);
  sf_prof_end(c, SF_K_NN_SELECT);
  SF_HIP(c, hipGetLastError());

  c->last_row_min.resize(n_l);
  c->last_row_arg.resize(n_l);
  SF_HIP(c, hipMemcpyAsync(c->last_row_min.data(), d_dist, (size_t)n_l * 8, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipMemcpyAsync(c->last_row_arg.data(), d_idx, (size_t)n_l * 4, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  }

  return sf_nn_walk_host(c, c->last_row_min.data(), c->last_row_arg.data(), n_l, n_r, c->params.netvlad_distance,
                         c->params.netvlad_max_matches_nb, out, cap, n_out);
}

// The NN kernels of this handle's local rows WITHOUT the walk and without a host round trip: d_row_min[n_local]
// (float64) and d_row_arg[n_local] (int32) in DEVICE memory, asynchronous on the handle's stream.  This is what one
// rank of the row-sharded NN stage contributes to the all-gather (data_handler.py:166-189 on
// a block of rows).  With nn_precision 1 the prefix filter runs at the ladder level the handle last settled on;
// d_status[0] = 1 reports a candidate set denser than the sparse limit (the minima are then undefined and the caller
// takes sf_nn_find_matches + sf_nn_last_row_minima, which walks the ladder), 0 otherwise.
// The filter path of the row minima in its two halves (the speculative step puts the verification of every candidate
// between them, on another stream): the stage-1 filter at the ladder level the handle last settled on ...
int sf_nn_filter_dev(sf_context* c, NnFilterOut* out) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  if (n_l <= 0 || n_r <= 0) return sf_fail(c, SF_EINVAL, "NN filter on an empty database (%d x %d)", n_l, n_r);
  int rc;
  if ((rc = nn_sync_masks(c)) != SF_OK) return rc;
  c->spec.valid = false;
  c->last_row_cand.clear();
  c->last_row_min.clear();        // (sf_nn_last_row_minima has nothing to report after this call)
  c->last_row_arg.clear();
  NnTrace tr;
  const int ld16 = (dim + 63) / 64 * 64;
  int levels[3], n_levels = 0;
  if (128 * 4 <= dim) levels[n_levels++] = 128;
  if (512 * 4 <= dim) levels[n_levels++] = 512;
  levels[n_levels++] = ld16;
  const int level = c->nn_force_full ? n_levels - 1 : std::min(std::max(c->nn_level, 0), n_levels - 1);
  NnFilterBufs fb;
  if ((rc = nn_filter_reserve(c, fb)) != SF_OK) return rc;
  if ((rc = nn_filter_launch(c, fb, level, levels[level], &tr)) != SF_OK) return rc;
  c->nn_last_kdims = levels[level];
  out->cand = fb.cand; out->count = fb.count; out->cdist = fb.cdist;
  out->limit = level < n_levels - 1 ? (unsigned)(8 * (size_t)n_l + 4096) : fb.cap;
  // the count stays on the device: the grids are sized for a typical sparse list and stride over a longer one
  out->typical = std::min<unsigned>(out->limit, (unsigned)(2 * (size_t)n_l + 1024));
  return SF_OK;
}

// ... and the exact re-evaluation of its candidates + their per-row minima, on the handle's CURRENT stream.
// d_row_cand / d_arg64 (both or neither): also the candidate-list index of each row's minimum.
int sf_nn_minima_of_candidates_dev(sf_context* c, const NnFilterOut& fo, double* d_row_min, int32_t* d_row_arg,
                                   int32_t* d_status, int32_t* d_row_cand, unsigned long long* d_arg64, bool throttle) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  const uint2* cand = (const uint2*)fo.cand;
  unsigned refine_wgs = (fo.typical + 3) / 4;
  if (throttle) {
    // beside a verification launch the re-evaluation has several times its own run time of slack: one workgroup per CU
    // walks the list (grid-stride) and leaves the HBM and the dispatcher to the verification kernel's first third
    if (c->n_cus <= 0) {
      int v = 0;
      c->n_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) ? v : 256;
    }
    refine_wgs = std::min(refine_wgs, (unsigned)c->n_cus);
  }
  sf_prof_begin(c, SF_K_NN_REFINE);
  hipLaunchKernelGGL(k_nn_refine, dim3(refine_wgs), dim3(256), 0, c->stream, cand, fo.count, fo.limit,
                     (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld, fo.cdist, 0u);
  sf_prof_end(c, SF_K_NN_REFINE);
  unsigned long long* mn = (unsigned long long*)d_row_min;
  const int wgs = (int)std::min<unsigned>((fo.typical + 255) / 256, 1024u);
  hipLaunchKernelGGL(k_nn_rowmin_init, dim3((n_l + 255) / 256), dim3(256), 0, c->stream, mn, d_row_arg, n_l, d_status, d_arg64);
  hipLaunchKernelGGL(k_nn_rowmin<0>, dim3(wgs), dim3(256), 0, c->stream, cand, fo.count, fo.limit, fo.cdist,
                     (const int*)c->d_ign_ptr.p, (const int*)c->d_ign_col.p, n_l, n_r, mn, d_row_arg, d_status, d_arg64);
  hipLaunchKernelGGL(k_nn_rowmin<1>, dim3(wgs), dim3(256), 0, c->stream, cand, fo.count, fo.limit, fo.cdist,
                     (const int*)c->d_ign_ptr.p, (const int*)c->d_ign_col.p, n_l, n_r, mn, d_row_arg, d_status, d_arg64);
  hipLaunchKernelGGL(k_nn_rowmin_finish, dim3((n_l + 255) / 256), dim3(256), 0, c->stream, d_row_arg, n_l,
                     (const unsigned long long*)d_arg64, d_row_cand);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

int sf_nn_row_minima_dev(sf_context* c, double* d_row_min, int32_t* d_row_arg, int32_t* d_status) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n, dim = c->nn_dim;
  if (n_l <= 0 || n_r <= 0) return sf_fail(c, SF_EINVAL, "sf_nn_row_minima_device on an empty database (%d x %d)", n_l, n_r);
  if (c->params.nn_precision != 0 && c->params.nn_precision != 1)
    return sf_fail(c, SF_EINVAL, "nn_precision %d unknown", c->params.nn_precision);
  const int ld = (dim + NN_BK - 1) / NN_BK * NN_BK;
  const int n_l_pad = (n_l + NN_BM - 1) / NN_BM * NN_BM, n_r_pad = (n_r + NN_BN - 1) / NN_BN * NN_BN;
  int rc;
  if (c->params.nn_precision == 1) {
    NnFilterOut fo;
    if ((rc = sf_nn_filter_dev(c, &fo)) != SF_OK) return rc;
    return sf_nn_minima_of_candidates_dev(c, fo, d_row_min, d_row_arg, d_status, nullptr, nullptr, false);
  }
  if ((rc = nn_sync_masks(c)) != SF_OK) return rc;
  c->spec.valid = false;
  c->last_row_cand.clear();
  c->last_row_min.clear();        // (sf_nn_last_row_minima has nothing to report after this call)
  c->last_row_arg.clear();
  // exact fp32-ranking path: its select kernel already leaves the minima in device memory
  const int n_strips = n_r_pad / 64;
  c->nn_last_kdims = 0;
  c->nn_coef_level = -1;
  const size_t part_bytes = (size_t)n_strips * n_l_pad * 16;
  if ((rc = sf_buf_reserve(c, c->nn_scalar, 64)) != SF_OK) return rc;
  unsigned* nb_max_bits = (unsigned*)((char*)c->nn_scalar.p + 16);
  SF_HIP(c, hipMemsetAsync(nb_max_bits, 0, 4, c->stream));
  SF_HIP(c, hipMemsetAsync(d_status, 0, 4, c->stream));
  if ((rc = sf_buf_reserve(c, c->nn_rowmin, part_bytes + (size_t)n_r_pad * 4)) != SF_OK) return rc;
  unsigned long long* part = (unsigned long long*)c->nn_rowmin.p;
  float* nb_eff = (float*)((char*)c->nn_rowmin.p + part_bytes);
  hipLaunchKernelGGL(k_nn_fill_norms, dim3((n_r_pad + 255) / 256), dim3(256), 0, c->stream, nb_eff,
                     (const float*)c->nn_recv.norms.p, (const uint8_t*)c->d_mask_other.p, n_r, n_r_pad, nb_max_bits);
  sf_prof_begin(c, SF_K_NN);
  hipLaunchKernelGGL(k_nn_argmin, dim3((n_r_pad / NN_BN) * (n_l_pad / NN_BM)), dim3(256), 0, c->stream,
                     (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p,
                     (const float*)c->nn_local.norms.p, nb_eff, (const int*)c->d_ign_ptr.p,
                     (const int*)c->d_ign_col.p, part, n_l_pad, ld, n_r_pad / NN_BN, n_l_pad / NN_BM);
  sf_prof_end(c, SF_K_NN);
  sf_prof_begin(c, SF_K_NN_SELECT);
  hipLaunchKernelGGL(k_nn_select, dim3((n_l + 3) / 4), dim3(256), 0, c->stream, part, n_strips, n_l, n_l_pad,
                     (const float*)c->nn_local.rows.p, (const float*)c->nn_recv.rows.p, dim, ld,
                     (const uint8_t*)c->d_mask_local.p, (const float*)c->nn_local.norms.p, nb_eff, nb_max_bits,
                     (const int*)c->d_ign_ptr.p, (const int*)c->d_ign_col.p, n_r, d_row_min, d_row_arg);
  sf_prof_end(c, SF_K_NN_SELECT);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
