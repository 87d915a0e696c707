// k_cnn.hip -- SURVEY.md section 8(f) rank 4, first half: NetVLAD descriptor inference (VGG16 trunk + NetVLAD layer +
// PCA whitening) on the device, the step in front of the NN search: DataHandler.compute_descriptors
// (data_handler.py:143-164) runs `nets.vgg16NetvladPca` (data_handler.py:63) on a batch of RGB keyframe images and
// keeps the first `netvlad_dimensions` values of every 4096-D row (data_handler.py:157-158).
//
// The network itself is third-party code the reference imports but does not vendor (netvlad_tf_open, python/netvlad_tf/
// nets.py + layers.py; weights from its checkpoint, not in the tree): PARITY UNPINNED.  Restated from its published
// definition, float32 throughout like the TensorFlow graph:
//   vgg16:   x - average_rgb;  conv1_1 relu conv1_2 pool relu | conv2_1 relu conv2_2 pool relu | conv3_1 relu conv3_2
//            relu conv3_3 pool relu | conv4_1 relu conv4_2 relu conv4_3 pool relu | conv5_1 relu conv5_2 relu conv5_3
//            (3 x 3, stride 1, 'same' zero padding, bias; 2 x 2 max pooling, stride 2, 'valid'; NO relu after conv5_3)
//   netVLAD: descriptor-wise L2 normalisation of conv5_3; soft assignment s = 1 x 1 conv to K = 64 clusters (no bias),
//            a = softmax(s); v[d][k] = sum_pixels a[p][k] (x[p][d] + C[d][k]); per-cluster normalisation
//            v / sqrt(sum_d v^2 + 1e-12) ("matconvnetNormalize"); flatten d-major, k-minor; the same normalisation of
//            the whole 32768-vector
//   WPCA:    1 x 1 conv = dense 32768 -> 4096 with bias, then tf.nn.l2_normalize (x / sqrt(max(sum x^2, 1e-12)))
//
// Layouts: activations NHWC float32 (one image at a time; a batch is a loop -- the layers are far past the size at
// which batching helps a 256-CU chip); conv weights [Cout][3][3][Cin] (= rows of a [Cout] x [9 Cin] matrix, K
// ordered tap-major), 1 x 1 weights [Cout][Cin], WPCA [4096][32768].  sf_netvlad_load documents the order it takes
// them in (TensorFlow's HWIO) and transposes on upload.
//
// Kernels:
//   k_conv3x3_first   Cin = 3 (27 MACs per output): direct, one thread per (pixel, 4 output channels)
//   k_conv_igemm_h    every other convolution as an implicit GEMM with SPLIT fp16 operands (x = hi + lo, three
//                     v_mfma_f32_32x32x16_f16 per product term: fp32-grade results, 3.1e-7 on the descriptor like the
//                     fp32 kernel, at a third of its matrix-pipe time); tile / tap split per layer MEASURED at the
//                     first inference of an image size (conv_autotune).  SF_CNN_FP32=1: k_conv_igemm below.
//   k_conv_igemm      the same on the fp32 matrix cores (v_mfma_f32_32x32x2f32, the
//                     128 x 128 x 32 tile of k_nn_argmin): M = pixels, N = Cout, K = taps x Cin; a 32-wide K step lies
//                     inside one tap (Cin is a multiple of 32), so the A tile of a step is 128 pixels x 128 B read
//                     straight from the NHWC activations of the shifted pixel (zero outside the image) -- no im2col
//                     buffer; bias + optional ReLU in the epilogue, 128-byte coalesced stores
//   k_pool2_relu      2 x 2 max pooling + ReLU
//   k_vlad_*          normalisation, softmax, aggregation (the assignment itself is k_conv_igemm with one tap)
//   k_wpca            one wavefront per output row of the 4096 x 32768 matrix (HBM-bound: 537 MB of weights per image)
// MFMA products are exact and accumulate in fp32 (the split drops only a_lo b_lo, 2^-22 of a product), so the result
// differs from a CPU fp32 evaluation essentially by summation order (tests: 1e-4 absolute on the unit-norm descriptor
// against a PyTorch fp32 CPU evaluation; measured 3.1e-7 for both kernels, tools/netvlad_error.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sf_internal.hpp"

namespace {

typedef float cf32x16 __attribute__((ext_vector_type(16)));
constexpr int CK = 32, CP = 36;    // K step 32 floats, LDS pitch 36 floats

// ---- packed split activations ("hl4") --------------------------------------------------------------------------
// Between the layers of the split-fp16 trunk an activation is stored ALREADY SPLIT: the four channels 4 g .. 4 g + 3 of a
// pixel occupy the 16 bytes a float4 would, as {hi[0..3], lo[0..3]} halves of x * CONV_ACT_SCALE (hi = fp16(x s),
// lo = fp16(x s - hi)).  Same strides as the fp32 tensor, so every index computation is unchanged; the consuming
// convolution's staging is then a copy (round 2 split every element again in every workgroup that read it -- 9 taps x
// Cout / TN times, 24 vector instructions per float4, more issue time than the MFMAs of a 64 x 64 tile's step).  The
// operands the matrix cores see are what they were: the producer does the split the consumer did; the 2 x 2 pooling
// takes the maximum of hi + lo (exact in fp32; the rounding is monotone, so it picks what max(x) would) and splits it
// again -- into the same pair except in rare ties at an fp16 rounding midpoint or binade edge, where it may choose the
// other 22-bit representation of the same value (descriptors of the two forms differ by <= 2e-8, a twentieth of their
// distance to a CPU fp32 evaluation).  conv5_3 (the VLAD layer's input) stays fp32.
// Measured on one box (tools/cnn_ab.sh): convolutions 10-17 % faster each, NetVLAD inference 1.34 -> 1.20 ms.
typedef _Float16 ch16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ch16x8 __attribute__((ext_vector_type(8)));
constexpr float CONV_ACT_SCALE = 0.0625f;

__device__ __forceinline__ void split4(const float4 v, float s, ch16x4& hi, ch16x4& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h = (_Float16)x[i];
    hi[i] = h;
    lo[i] = (_Float16)(x[i] - (float)h);
  }
}
// (all in registers: a union or an indexed vector here becomes a private array that the compiler moves to LDS -- 2 KB per
//  workgroup and an LDS round trip per element, which tripled the run time of the pooling kernel)
typedef _Float16 ch16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float hl_word(_Float16 a, _Float16 b) {
  const ch16x2 v = {a, b};
  return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float4 hl4_repack(const float4 x) {          // x * CONV_ACT_SCALE -> packed
  const _Float16 h0 = (_Float16)x.x, h1 = (_Float16)x.y, h2 = (_Float16)x.z, h3 = (_Float16)x.w;
  const _Float16 l0 = (_Float16)(x.x - (float)h0), l1 = (_Float16)(x.y - (float)h1), l2 = (_Float16)(x.z - (float)h2),
                 l3 = (_Float16)(x.w - (float)h3);
  return make_float4(hl_word(h0, h1), hl_word(h2, h3), hl_word(l0, l1), hl_word(l2, l3));
}
__device__ __forceinline__ float4 hl4_pack(const float4 v) {             // true-scale values -> packed
  return hl4_repack(make_float4(v.x * CONV_ACT_SCALE, v.y * CONV_ACT_SCALE, v.z * CONV_ACT_SCALE, v.w * CONV_ACT_SCALE));
}
__device__ __forceinline__ float4 hl4_values(const float4 p) {           // packed -> hi + lo (x * CONV_ACT_SCALE, exact sums)
  const ch16x2 h01 = __builtin_bit_cast(ch16x2, p.x), h23 = __builtin_bit_cast(ch16x2, p.y);
  const ch16x2 l01 = __builtin_bit_cast(ch16x2, p.z), l23 = __builtin_bit_cast(ch16x2, p.w);
  return make_float4((float)h01[0] + (float)l01[0], (float)h01[1] + (float)l01[1], (float)h23[0] + (float)l23[0],
                     (float)h23[1] + (float)l23[1]);
}

// in: [H][W][3] float, wgt: [64][27] (tap-major, channel-minor), out: [H][W][64] (fp32, or packed split halves when `packed`); mean subtraction fused
__global__ void __launch_bounds__(256)
k_conv3x3_first(const float* __restrict__ in, int H, int W, const float* __restrict__ wgt, const float* __restrict__ bias,
                const float* __restrict__ mean, float* __restrict__ out, int relu, int packed, int Himg) {
  // (H = rows of the whole input, Himg = rows of ONE image: a batch is a stack of images, a tap never crosses into the next)
  __shared__ __attribute__((aligned(16))) float sw[64 * 27];
  __shared__ __attribute__((aligned(16))) float sb[64];
  // (tap-major in LDS: sw[k][channel], so that four channels' weights of a tap come in one 16-byte broadcast read)
  for (int i = threadIdx.x; i < 64 * 27; i += 256) sw[(i % 27) * 64 + i / 27] = wgt[i];
  if (threadIdx.x < 64) sb[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int g = blockIdx.x * 256 + threadIdx.x;        // (pixel, group of 16 output channels)
  const int p = g >> 2, c16 = (g & 3) * 16;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W, yi = y % Himg;
  float v[27];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1, yyi = yi + t / 3 - 1;
    const bool ok = yyi >= 0 && yyi < Himg && xx >= 0 && xx < W;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[3 * t + c] = ok ? in[((size_t)yy * W + xx) * 3 + c] - mean[c] : 0.f;
  }
  // (four threads per pixel: the 27 inputs are fetched 4 times instead of 16, and the four write one 256-byte row)
#pragma unroll
  for (int c4 = 0; c4 < 16; c4 += 4) {
    float4 a = *reinterpret_cast<const float4*>(&sb[c16 + c4]);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float4 w4 = *reinterpret_cast<const float4*>(&sw[k * 64 + c16 + c4]);
      a.x = fmaf(v[k], w4.x, a.x); a.y = fmaf(v[k], w4.y, a.y); a.z = fmaf(v[k], w4.z, a.z); a.w = fmaf(v[k], w4.w, a.w);
    }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    *reinterpret_cast<float4*>(out + (size_t)p * 64 + c16 + c4) = packed ? hl4_pack(a) : a;
  }
}

// Implicit GEMM: out[p][n] = act(bias[n] + sum_{t, c} in[shift_t(p)][c] * wgt[n][t * Cin + c]).
// TAPS = 9 (3 x 3, 'same') or 1 (1 x 1).  P = H * W pixels, Cin % 32 == 0.  Tile TM pixels x TN output channels
// (64 or 128 each; a wavefront owns a (TM / 2) x (TN / 2) quarter): the late VGG layers have few pixels (40 x 30 at
// conv5 for a 640 x 480 image) and take the 64-wide tiles so that the grid still covers the chip.  The operand tiles
// of K step s + 1 are requested into registers before the MFMAs of step s and written to LDS after them.
template <int TAPS, int TM, int TN>
__global__ void __launch_bounds__(256)
k_conv_igemm(const float* __restrict__ in, int H, int W, int Cin, const float* __restrict__ wgt, int Cout,
             const float* __restrict__ bias, float* __restrict__ out, int relu) {
  constexpr int NI = TM / 64, NJ = TN / 64, QA = TM / 32, QB = TN / 32;
  __shared__ __attribute__((aligned(16))) float sA[TM * CP];
  __shared__ __attribute__((aligned(16))) float sB[TN * CP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int P = H * W;
  const int row0 = blockIdx.x * TM, col0 = blockIdx.y * TN;
  const int K = TAPS * Cin;
  // split K (gridDim.z > 1): slice z contracts taps [z TAPS / Z, (z + 1) TAPS / Z) into its own partial output
  // [z][P][Cout]; bias and ReLU are then applied by k_sum_partials (a fixed summation order: deterministic)
  const int k_lo = (int)((blockIdx.z * TAPS) / gridDim.z) * Cin, k_hi = (int)(((blockIdx.z + 1) * TAPS) / gridDim.z) * Cin;
  out += (size_t)blockIdx.z * P * Cout;
  cf32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int srow = tid >> 3, sk4 = (tid & 7) * 4;     // staging: 32 rows x 8 float4 per pass
  int py[QA], px[QA];
  bool pv[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int p = row0 + srow + 32 * q;
    pv[q] = p < P;
    py[q] = pv[q] ? p / W : 0;
    px[q] = pv[q] ? p - py[q] * W : 0;
  }
  float4 ra[QA], rb[QB];
  auto fetch = [&](int k0) {
    const int tap = TAPS == 1 ? 0 : k0 / Cin, c0 = k0 - tap * Cin;
    const int dy = TAPS == 1 ? 0 : tap / 3 - 1, dx = TAPS == 1 ? 0 : tap % 3 - 1;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int yy = py[q] + dy, xx = px[q] + dx;
      ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pv[q] && yy >= 0 && yy < H && xx >= 0 && xx < W)
        ra[q] = *reinterpret_cast<const float4*>(in + ((size_t)yy * W + xx) * Cin + c0 + sk4);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col0 + srow + 32 * q < Cout)
        rb[q] = *reinterpret_cast<const float4*>(wgt + (size_t)(col0 + srow + 32 * q) * K + k0 + sk4);
    }
  };
  fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += CK) {
    __syncthreads();                                  // the previous step's LDS reads are complete
#pragma unroll
    for (int q = 0; q < QA; ++q) *reinterpret_cast<float4*>(&sA[(srow + 32 * q) * CP + sk4]) = ra[q];
#pragma unroll
    for (int q = 0; q < QB; ++q) *reinterpret_cast<float4*>(&sB[(srow + 32 * q) * CP + sk4]) = rb[q];
    __syncthreads();
    if (k0 + CK < k_hi) fetch(k0 + CK);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        a[i] = *reinterpret_cast<const float4*>(&sA[((TM / 2) * wr + 32 * i + l31) * CP + 8 * q + 4 * h]);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        b[j] = *reinterpret_cast<const float4*>(&sB[((TN / 2) * wc + 32 * j + l31) * CP + 8 * q + 4 * h]);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }
  // C layout: column (= output channel) = lane & 31, row (= pixel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = col0 + (TN / 2) * wc + 32 * j + l31;
    if (n >= Cout) continue;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = row0 + (TM / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (p < P) {
          const float v = acc[i][j][r] + bn;
          out[(size_t)p * Cout + n] = relu ? fmaxf(v, 0.f) : v;
        }
      }
  }
}

// The same implicit GEMM with SPLIT fp16 operands on the fp16 matrix cores: x = hi + lo with hi = fp16(x),
// lo = fp16(x - hi) -- 22 significant bits -- and
//   a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi   (the dropped a_lo b_lo is 2^-22 of the product);
// products are exact in the fp32 accumulator.  The MFMA is CDNA4's v_mfma_f32_32x32x16_f16: 16 k per instruction in the
// 32 cycles the CDNA3-era v_mfma_f32_32x32x8_f16 takes for 8 (tools/ubench/mfma_f16_rate.hip: 32.2 against 32.4 cycles
// -- the old opcode runs at HALF the fp16 rate of the chip; rounds 2 built this kernel on it).  The
// weights are split once at load time (and scaled by a power of two so that the largest is about 8: small weights keep
// their low part out of the subnormals), the activations when a tile is staged into LDS (scaled by 2^-4: headroom for
// activations up to 10^6); both scalings are undone exactly in the epilogue.  The lane layout is the fp32 kernel's:
// a lane's two float4 of the fp32 form (k = 8 h .. 8 h + 7 of a 16-wide slice) are the 8-half operand of ONE x16 MFMA.
constexpr int CONV_ZERO_PAGE = 1024;   // floats of zeros the host keeps in front of every activation buffer

// CKT = channels of a tap per K step (32 or 64; 64: half the barriers per MFMA, twice the LDS -- the choice is measured per
// layer with the tile, it does not change the order of a pixel's sum)
template <int TM, int TN, int CKT>
__global__ void __launch_bounds__(256)
k_conv_igemm_h(const float* __restrict__ in, int H, int W, int Cin, const _Float16* __restrict__ wgt_hi,
               const _Float16* __restrict__ wgt_lo, float out_scale, int Cout, const float* __restrict__ bias,
               float* __restrict__ out, int relu, int out_packed, int Himg) {
  // `in`: packed split activations (above); `out`: fp32, or packed when out_packed; H rows = a stack of images of Himg rows
  constexpr int TAPS = 9;
  constexpr int GK = CKT / 4, RP = 256 / GK;            // staging: RP rows x GK groups of 4 k per pass
  constexpr int CPH = CKT + 8;                          // LDS pitch in halves (80 / 144 bytes: conflict-free ds_read_b128)
  constexpr int NI = TM / 64, NJ = TN / 64, QA = TM / RP, QB = TN / RP;
  __shared__ __attribute__((aligned(16))) _Float16 sAh[TM * CPH];
  __shared__ __attribute__((aligned(16))) _Float16 sAl[TM * CPH];
  __shared__ __attribute__((aligned(16))) _Float16 sBh[TN * CPH];
  __shared__ __attribute__((aligned(16))) _Float16 sBl[TN * CPH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int P = H * W;
  const int row0 = blockIdx.x * TM, col0 = blockIdx.y * TN;
  const int K = TAPS * Cin;
  const int k_lo = (int)((blockIdx.z * TAPS) / gridDim.z) * Cin, k_hi = (int)(((blockIdx.z + 1) * TAPS) / gridDim.z) * Cin;
  out += (size_t)blockIdx.z * P * Cout;
  cf32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int srow = tid / GK, sk4 = (tid % GK) * 4;
  // Operand addresses of a K step (one tap, 32 channels) = a UNIFORM base that moves with the step + a 32-bit lane offset
  // that does not: global_load saddr + voffset, no 64-bit multiplies, no divisions, no branches in the loop (round 2
  // recomputed tap = k0 / Cin, the shifted pixel's address and a bounds test per row and step, and branched around
  // every load: ~100 vector instructions per step beside 12-24 MFMAs).  A tap that falls outside the image (or a row
  // past the last pixel) reads ZEROS from the 4 KB page the host keeps in front of every activation buffer
  // (CONV_ZERO_PAGE floats): its lane offset is swapped for one inside that page, the data needs no masking.
  unsigned oa[QA], vmask[QA];                        // byte offset of the row's centre pixel from in - page; valid taps
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int p = row0 + srow + RP * q;
    const bool pv = p < P;
    const int py = pv ? p / W : 0, px = pv ? p - py * W : 0, pyi = py % Himg;
    oa[q] = (unsigned)(CONV_ZERO_PAGE * 4) + ((unsigned)(py * W + px) * (unsigned)Cin + (unsigned)sk4) * 4u;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = pyi + t / 3 - 1, xx = px + t % 3 - 1;          // (inside the row's own image)
      mk |= (pv && yy >= 0 && yy < Himg && xx >= 0 && xx < W) ? (1u << t) : 0u;
    }
    vmask[q] = mk;
  }
  const unsigned zoff = (unsigned)sk4 * 4u;           // (16-byte aligned, inside the zero page)
  unsigned ob[QB];                                    // byte offset of the weight row (clamped: columns >= Cout are never stored)
#pragma unroll
  for (int q = 0; q < QB; ++q) ob[q] = ((unsigned)min(col0 + srow + RP * q, Cout - 1) * (unsigned)K + (unsigned)sk4) * 2u;
  const char* in_page = reinterpret_cast<const char*>(in) - CONV_ZERO_PAGE * 4;
  float4 ra[QA];
  ch16x4 rbh[QB], rbl[QB];
  int f_tap = k_lo / Cin, f_c0 = 0;                   // the step the NEXT fetch loads (k_lo is a multiple of Cin)
  auto fetch = [&]() {
    const int dy = ((f_tap * 11) >> 5) - 1, dx = f_tap - 3 * (dy + 1) - 1;      // tap / 3 - 1, tap % 3 - 1 for tap < 9
    const unsigned step = (unsigned)(((dy * W + dx) * Cin + f_c0) * 4);          // (wraps; offset + step >= 0 where valid)
    const unsigned bit = 1u << f_tap;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const unsigned off = (vmask[q] & bit) ? oa[q] + step : zoff;
      ra[q] = *reinterpret_cast<const float4*>(in_page + off);
    }
    const size_t kb = (size_t)(f_tap * Cin + f_c0) * 2;
    const char* bh = reinterpret_cast<const char*>(wgt_hi) + kb;
    const char* bl = reinterpret_cast<const char*>(wgt_lo) + kb;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      rbh[q] = *reinterpret_cast<const ch16x4*>(bh + ob[q]);
      rbl[q] = *reinterpret_cast<const ch16x4*>(bl + ob[q]);
    }
    f_c0 += CKT;
    if (f_c0 == Cin) { f_c0 = 0; ++f_tap; }
  };
  fetch();
  for (int k0 = k_lo; k0 < k_hi; k0 += CKT) {
    __syncthreads();                                  // the previous step's LDS reads are complete
#pragma unroll
    for (int q = 0; q < QA; ++q) {   // the activations arrive split: a copy (words 0, 1 = hi, words 2, 3 = lo)
      *reinterpret_cast<float2*>(&sAh[(srow + RP * q) * CPH + sk4]) = make_float2(ra[q].x, ra[q].y);
      *reinterpret_cast<float2*>(&sAl[(srow + RP * q) * CPH + sk4]) = make_float2(ra[q].z, ra[q].w);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      *reinterpret_cast<ch16x4*>(&sBh[(srow + RP * q) * CPH + sk4]) = rbh[q];
      *reinterpret_cast<ch16x4*>(&sBl[(srow + RP * q) * CPH + sk4]) = rbl[q];
    }
    __syncthreads();
    if (k0 + CKT < k_hi) fetch();
#pragma unroll
    for (int q = 0; q < CKT / 16; ++q) {   // 16 of the step's k per MFMA: lane (l31, h) holds k = 16 q + 8 h .. + 7
      ch16x8 ah[NI], al[NI], bh[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int o = ((TM / 2) * wr + 32 * i + l31) * CPH + 16 * q + 8 * h;
        ah[i] = *reinterpret_cast<const ch16x8*>(&sAh[o]);
        al[i] = *reinterpret_cast<const ch16x8*>(&sAl[o]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int o = ((TN / 2) * wc + 32 * j + l31) * CPH + 16 * q + 8 * h;
        bh[j] = *reinterpret_cast<const ch16x8*>(&sBh[o]);
        bl[j] = *reinterpret_cast<const ch16x8*>(&sBl[o]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // Packed output: a lane holds ONE channel of 16 pixels; the four channels of a 16-byte group sit in the four lanes of a
  // quad.  Each lane splits its value into P = hi | lo << 16, takes P of two quad neighbours (DPP quad permutes) and
  // assembles the 32-bit word of the group that lands at ITS channel's address: words 0 / 1 = hi of channels (0, 1) /
  // (2, 3), words 2 / 3 = lo of the same -- the store pattern is the fp32 one.
  const int kq = l31 & 3;
  const unsigned perm_sel = kq < 2 ? 0x05040100u : 0x07060302u;     // low halves of (Pa, Pb) / high halves
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = col0 + (TN / 2) * wc + 32 * j + l31;
    const bool n_ok = n < Cout;                                      // (uniform over a quad: Cout % 4 == 0)
    const float bn = (bias && n_ok) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = row0 + (TM / 2) * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][j][r] * out_scale + bn;
        v = relu ? fmaxf(v, 0.f) : v;
        if (out_packed) {
          const float xs = v * CONV_ACT_SCALE;
          const _Float16 hh = (_Float16)xs;
          const _Float16 ll = (_Float16)(xs - (float)hh);
          unsigned short hb, lb;
          __builtin_memcpy(&hb, &hh, 2);
          __builtin_memcpy(&lb, &ll, 2);
          const int Pw = (int)((unsigned)hb | ((unsigned)lb << 16));
          // quad_perm [0, 2, 0, 2] and [1, 3, 1, 3]: lanes 0 / 2 of a quad read lanes (0, 1), lanes 1 / 3 read (2, 3)
          const unsigned Pa = (unsigned)__builtin_amdgcn_update_dpp(0, Pw, 0x88, 0xF, 0xF, false);
          const unsigned Pb = (unsigned)__builtin_amdgcn_update_dpp(0, Pw, 0xDD, 0xF, 0xF, false);
          const unsigned word = __builtin_amdgcn_perm(Pb, Pa, perm_sel);
          if (p < P && n_ok) reinterpret_cast<unsigned*>(out)[(size_t)p * Cout + n] = word;
        } else if (p < P && n_ok) {
          out[(size_t)p * Cout + n] = v;
        }
      }
  }
}

// 2 x 2 max pooling (stride 2, 'valid') + ReLU; in [H][W][C] -> out [H/2][W/2][C], 4 channels per thread
__global__ void __launch_bounds__(256)
k_pool2_relu(const float* __restrict__ in, int H, int W, int C, float* __restrict__ out, int packed) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)Ho * Wo * C4) return;
  const int c4 = (int)(g % C4);
  const size_t p = g / C4;
  const int xo = (int)(p % Wo), yo = (int)(p / Wo);
  const float4* src = reinterpret_cast<const float4*>(in);
  float4 a = src[((size_t)(2 * yo) * W + 2 * xo) * C4 + c4], b = src[((size_t)(2 * yo) * W + 2 * xo + 1) * C4 + c4];
  float4 c = src[((size_t)(2 * yo + 1) * W + 2 * xo) * C4 + c4], d = src[((size_t)(2 * yo + 1) * W + 2 * xo + 1) * C4 + c4];
  if (packed) { a = hl4_values(a); b = hl4_values(b); c = hl4_values(c); d = hl4_values(d); }   // (scaled: max and ReLU do not care)
  float4 m;
  m.x = fmaxf(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), 0.f);
  m.y = fmaxf(fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)), 0.f);
  m.z = fmaxf(fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), 0.f);
  m.w = fmaxf(fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)), 0.f);
  reinterpret_cast<float4*>(out)[g] = packed ? hl4_repack(m) : m;
}

// tf.nn.l2_normalize over the channels of every pixel: x / sqrt(max(sum x^2, 1e-12)); one wavefront per pixel
__global__ void __launch_bounds__(256)
k_l2norm_rows(float* __restrict__ x, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = x + (size_t)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(p[c], p[c], s);
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float inv = 1.0f / sqrtf(fmaxf(s, 1e-12f));
  for (int c = lane; c < C; c += 64) p[c] *= inv;
}

// out[i] = act(bias[i % C] + (part[0][i] + part[1][i]) + ... ), 4 values per thread
__global__ void __launch_bounds__(256)
k_sum_partials(const float* __restrict__ part, int S, size_t n, int C, const float* __restrict__ bias, float* __restrict__ out,
               int relu, int packed) {
  const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (g >= n) return;
  float4 a = *reinterpret_cast<const float4*>(part + g);
  for (int z = 1; z < S; ++z) {
    const float4 q = *reinterpret_cast<const float4*>(part + (size_t)z * n + g);
    a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
  }
  const float4 bz = *reinterpret_cast<const float4*>(bias + (g % C));
  a.x += bz.x; a.y += bz.y; a.z += bz.z; a.w += bz.w;
  if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  *reinterpret_cast<float4*>(out + g) = packed ? hl4_pack(a) : a;
}

// softmax over the K clusters of every pixel (in place); one wavefront per pixel, lane = cluster, K <= 64
__global__ void __launch_bounds__(256)
k_softmax_rows(float* __restrict__ s, int rows, int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = s + (size_t)row * K;
  const float v = lane < K ? p[lane] : -INFINITY;
  float m = v;
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  const float e = lane < K ? expf(v - m) : 0.f;
  float sum = e;
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane < K) p[lane] = e / sum;
}

// v[d][k] = sum_p a[p][k] (x[p][d] + C[d][k]); one workgroup per 2 consecutive d (lane = cluster, the four wavefronts
// split the pixels; a[p][:] 256 bytes per wavefront), K <= 64
__global__ void __launch_bounds__(256)
k_vlad_aggregate(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ centers, int P,
                 int D, int K, float* __restrict__ v) {
  // (two d per workgroup: 256 workgroups for D = 512, one per CU; the pixel loop is unrolled so that eight pixels'
  //  loads are in flight per wavefront -- it was one dependent round trip per pixel, 75 us for 1 200 pixels)
  __shared__ float red[4][2][64];
  const int d0 = blockIdx.x * 2, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[2] = {0.f, 0.f};
  if (lane < K) {
    float c[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) c[q] = centers[(size_t)(d0 + q) * K + lane];
#pragma unroll 8
    for (int p = wave; p < P; p += 4) {
      const float ap = a[(size_t)p * K + lane];
      const float2 xv = *reinterpret_cast<const float2*>(x + (size_t)p * D + d0);
      acc[0] = fmaf(ap, xv.x + c[0], acc[0]);
      acc[1] = fmaf(ap, xv.y + c[1], acc[1]);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) red[wave][q][lane] = acc[q];
  __syncthreads();
  if (lane < K && wave < 2) {
    const int q = wave;          // wavefront q finishes d0 + q
    v[(size_t)(d0 + q) * K + lane] = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
  }
}

// matconvnetNormalize per cluster (over d) of v[d][k]: inv[k] = 1 / sqrt(sum_d v^2 + 1e-12); one workgroup per cluster
__global__ void __launch_bounds__(256)
k_vlad_cluster_norms(const float* __restrict__ v, int D, int K, float* __restrict__ inv) {
  __shared__ float red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int d = tid; d < D; d += 256) { const float t = v[(size_t)d * K + k]; s = fmaf(t, t, s); }
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) inv[k] = 1.0f / sqrtf(red[0] + 1e-12f);
}

// v <- v * inv[k], then matconvnetNormalize of the whole vector; one workgroup of 1024 threads
__global__ void __launch_bounds__(1024)
k_vlad_normalize(float* __restrict__ v, int D, int K, const float* __restrict__ inv_k) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  float tot = 0.f;
  for (int i = tid; i < D * K; i += 1024) {
    const float t = v[i] * inv_k[i % K];
    v[i] = t;
    tot = fmaf(t, t, tot);
  }
  red[tid] = tot;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float inv = 1.0f / sqrtf(red[0] + 1e-12f);
  for (int i = tid; i < D * K; i += 1024) v[i] *= inv;
}

// y[r] = b[r] + W[r][:] . v ; one wavefront per row
__global__ void __launch_bounds__(256)
k_wpca(const float* __restrict__ Wm, const float* __restrict__ b, const float* __restrict__ v, int rows, int cols,
       float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* w4 = reinterpret_cast<const float4*>(Wm + (size_t)row * cols);
  const float4* v4 = reinterpret_cast<const float4*>(v);
  float s = 0.f;
  for (int c = lane; c < cols / 4; c += 64) {
    const float4 a = w4[c], q = v4[c];
    s = fmaf(a.x, q.x, s); s = fmaf(a.y, q.y, s); s = fmaf(a.z, q.z, s); s = fmaf(a.w, q.w, s);
  }
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) y[row] = s + b[row];
}

// tf.nn.l2_normalize of one vector, writing the first n_out values; one workgroup
// The same product for B <= 4 VLAD vectors of a batch (data_handler.py:149-156 infers up to netvlad_batch_size = 3 images
// per call): one pass over the 537 MB of weights serves them all; per vector the arithmetic is k_wpca's, term by term,
// so a batch gives the bits of the single-image calls.
template <int B>
__global__ void __launch_bounds__(256)
k_wpca_batch(const float* __restrict__ Wm, const float* __restrict__ b, const float* __restrict__ v, int rows, int cols,
             float* __restrict__ y, int y_pitch) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* w4 = reinterpret_cast<const float4*>(Wm + (size_t)row * cols);
  float s[B];
#pragma unroll
  for (int k = 0; k < B; ++k) s[k] = 0.f;
  for (int c = lane; c < cols / 4; c += 64) {
    const float4 a = w4[c];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const float4 q = reinterpret_cast<const float4*>(v + (size_t)k * cols)[c];
      s[k] = fmaf(a.x, q.x, s[k]); s[k] = fmaf(a.y, q.y, s[k]); s[k] = fmaf(a.z, q.z, s[k]); s[k] = fmaf(a.w, q.w, s[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < B; ++k) {
    float t = s[k];
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) y[(size_t)k * y_pitch + row] = t + b[row];
  }
}

__global__ void __launch_bounds__(256)
k_l2norm_vec(const float* __restrict__ y, int n, float* __restrict__ out, int n_out) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < n; i += 256) s = fmaf(y[i], y[i], s);
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float inv = 1.0f / sqrtf(fmaxf(red[0], 1e-12f));
  for (int i = tid; i < n_out; i += 256) out[i] = y[i] * inv;
}

const int VGG_COUT[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int VGG_CIN[13] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
const bool VGG_RELU[13] = {true, false, true, false, true, true, false, true, true, false, true, true, false};
const bool VGG_POOL[13] = {false, true, false, true, false, false, true, false, false, true, false, false, false};

}  // namespace

struct ConvCfg { int tm, tn, split, ck; };   // ck: channels per K step of the split-fp16 kernel (32 / 64)

struct sf_netvlad_model {
  Buf conv_w[13], conv_b[13], mean, assign_w, centers, pca_w, pca_b;
  Buf conv_wh[13], conv_wl[13];     // fp16 high / low parts of the scaled weights (k_conv_igemm_h)
  float conv_out_scale[13] = {};    // 1 / (weight scale x activation scale): exact powers of two
  bool split_f16 = true;            // SF_CNN_FP32=1: the fp32 matrix-core kernels
  const void* act_zeroed[2] = {nullptr, nullptr};   // the allocations whose leading zero page has been written
  size_t act_zeroed_bytes[2] = {0, 0};
  struct Tune { ConvCfg cfg[13] = {}; int h = 0, w = 0, n = 0; };   // tile / K step / split of every layer, measured for
  Tune tune[2];                     // ... an image size and a stack of n images (conv_autotune): [0] n = 1, [1] the last n > 1
  Buf act[2], vlad, pca_y, partial;
  int clusters = 0, pca_dim = 0;
};

void sf_netvlad_free(sf_context* c) {
  if (!c->netvlad) return;
  sf_netvlad_model* m = c->netvlad;
  for (int i = 0; i < 13; ++i) { sf_buf_free(m->conv_w[i]); sf_buf_free(m->conv_b[i]); sf_buf_free(m->conv_wh[i]); sf_buf_free(m->conv_wl[i]); }
  Buf* bs[] = {&m->mean, &m->assign_w, &m->centers, &m->pca_w, &m->pca_b, &m->act[0], &m->act[1], &m->vlad, &m->pca_y, &m->partial};
  for (Buf* b : bs) sf_buf_free(*b);
  delete m;
  c->netvlad = nullptr;
}

static int upload(sf_context* c, Buf& b, const float* host, size_t n) {
  int rc = sf_buf_reserve(c, b, n * sizeof(float));
  if (rc != SF_OK) return rc;
  SF_HIP(c, hipMemcpyAsync(b.p, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return SF_OK;
}

// Host weights in TensorFlow's layouts: conv kernels HWIO [3][3][Cin][Cout], assignment [1][1][512][K] = [512][K],
// cluster centers [512][K], WPCA kernel [1][1][512 K][pca_dim] = [512 K][pca_dim].  Transposed here to output-major.
int sf_netvlad_load_impl(sf_context* c, const sf_netvlad_weights* w) {
  if (!w || !w->average_rgb || !w->assignment || !w->cluster_centers || !w->wpca_kernel || !w->wpca_bias)
    return sf_fail(c, SF_EINVAL, "NetVLAD weights missing");
  if (w->clusters < 1 || w->clusters > 64 || w->pca_dim < 1)
    return sf_fail(c, SF_ERANGE, "NetVLAD: 1..64 clusters and a positive WPCA dimension, not %d / %d", w->clusters, w->pca_dim);
  for (int i = 0; i < 13; ++i)
    if (!w->conv_kernel[i] || !w->conv_bias[i]) return sf_fail(c, SF_EINVAL, "VGG16 convolution %d missing", i);
  sf_netvlad_free(c);
  sf_netvlad_model* m = new (std::nothrow) sf_netvlad_model();
  if (!m) return sf_fail(c, SF_ENOMEM, "out of host memory");
  c->netvlad = m;
  m->split_f16 = getenv("SF_CNN_FP32") == nullptr;
  m->clusters = w->clusters;
  m->pca_dim = w->pca_dim;
  int rc;
  std::vector<float> t;
  for (int i = 0; i < 13; ++i) {
    const int ci = VGG_CIN[i], co = VGG_COUT[i];
    t.resize((size_t)co * 9 * ci);
    for (int tap = 0; tap < 9; ++tap)
      for (int a = 0; a < ci; ++a)
        for (int o = 0; o < co; ++o) t[((size_t)o * 9 + tap) * ci + a] = w->conv_kernel[i][((size_t)tap * ci + a) * co + o];
    if ((rc = upload(c, m->conv_w[i], t.data(), t.size())) != SF_OK) return rc;
    SF_HIP(c, hipStreamSynchronize(c->stream));          // (t is reused)
    if (i > 0) {
      // split once: w 2^e = hi + lo in fp16, 2^e chosen so that the largest |w| lands in [8, 16)
      float wmax = 0.f;
      for (float v : t) wmax = std::max(wmax, std::fabs(v));
      int e = 0;
      if (wmax > 0.f && std::isfinite(wmax)) e = 3 - (int)std::floor(std::log2(wmax));
      e = std::max(-20, std::min(20, e));
      const float ws = std::ldexp(1.0f, e);
      std::vector<_Float16> th(t.size()), tl(t.size());
      for (size_t k = 0; k < t.size(); ++k) {
        const float x = t[k] * ws;
        const _Float16 hh = (_Float16)x;
        th[k] = hh;
        tl[k] = (_Float16)(x - (float)hh);
      }
      if ((rc = sf_buf_reserve(c, m->conv_wh[i], t.size() * 2)) != SF_OK) return rc;
      if ((rc = sf_buf_reserve(c, m->conv_wl[i], t.size() * 2)) != SF_OK) return rc;
      SF_HIP(c, hipMemcpyAsync(m->conv_wh[i].p, th.data(), t.size() * 2, hipMemcpyHostToDevice, c->stream));
      SF_HIP(c, hipMemcpyAsync(m->conv_wl[i].p, tl.data(), t.size() * 2, hipMemcpyHostToDevice, c->stream));
      SF_HIP(c, hipStreamSynchronize(c->stream));
      m->conv_out_scale[i] = std::ldexp(1.0f, -e) / CONV_ACT_SCALE;
    }
    if ((rc = upload(c, m->conv_b[i], w->conv_bias[i], co)) != SF_OK) return rc;
  }
  if ((rc = upload(c, m->mean, w->average_rgb, 3)) != SF_OK) return rc;
  const int K = w->clusters, D = 512;
  t.resize((size_t)K * D);
  for (int d = 0; d < D; ++d)
    for (int k = 0; k < K; ++k) t[(size_t)k * D + d] = w->assignment[(size_t)d * K + k];
  // (the assignment runs on the 128-wide tile: its K <= 64 rows are padded with zero rows by the kernel's guard)
  if ((rc = upload(c, m->assign_w, t.data(), t.size())) != SF_OK) return rc;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if ((rc = upload(c, m->centers, w->cluster_centers, (size_t)D * K)) != SF_OK) return rc;
  const size_t cols = (size_t)D * K;
  if ((rc = sf_buf_reserve(c, m->pca_w, cols * w->pca_dim * sizeof(float))) != SF_OK) return rc;
  {
    // [cols][pca_dim] -> [pca_dim][cols], in slabs of rows so that the host copy stays small
    const int slab = 64;
    t.resize((size_t)slab * cols);
    for (int r0 = 0; r0 < w->pca_dim; r0 += slab) {
      const int nr = std::min(slab, w->pca_dim - r0);
      for (size_t cidx = 0; cidx < cols; ++cidx)
        for (int r = 0; r < nr; ++r) t[(size_t)r * cols + cidx] = w->wpca_kernel[cidx * w->pca_dim + r0 + r];
      SF_HIP(c, hipMemcpyAsync((float*)m->pca_w.p + (size_t)r0 * cols, t.data(), (size_t)nr * cols * sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
      SF_HIP(c, hipStreamSynchronize(c->stream));
    }
  }
  if ((rc = upload(c, m->pca_b, w->wpca_bias, w->pca_dim)) != SF_OK) return rc;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  return SF_OK;
}

// One 3 x 3 convolution layer (i >= 1) in a given configuration: tile tm x tn, S-way tap split (S > 1: partial outputs,
// summed -- with bias and ReLU -- by k_sum_partials in a fixed order).
static int conv_layer(sf_context* c, sf_netvlad_model* m, int i, const float* src, int h, int w, float* dst, ConvCfg cfg,
                      int himg) {   // h = rows of the stack, himg = rows of one image at this layer
  // split-fp16 trunk: activations travel packed (hl4) except the last layer's output, the VLAD head's fp32 input
  const int pk = (m->split_f16 && i != 12) ? 1 : 0;
  const int P = h * w, co = VGG_COUT[i], S = cfg.split, tm = cfg.tm, tn = cfg.tn;
  int rc;
  float* cdst = dst;
  if (S > 1) {
    if ((rc = sf_buf_reserve(c, m->partial, (size_t)S * P * co * sizeof(float))) != SF_OK) return rc;
    cdst = (float*)m->partial.p;
  }
#define SF_CONV_H(TM_, TN_, CK_)                                                                                    \
  hipLaunchKernelGGL((k_conv_igemm_h<TM_, TN_, CK_>), dim3((P + TM_ - 1) / TM_, (co + TN_ - 1) / TN_, S), dim3(256), 0, \
                     c->stream, src, h, w, VGG_CIN[i], (const _Float16*)m->conv_wh[i].p,                              \
                     (const _Float16*)m->conv_wl[i].p, m->conv_out_scale[i], co,                                      \
                     S > 1 ? (const float*)nullptr : (const float*)m->conv_b[i].p, cdst,                              \
                     S > 1 ? 0 : (VGG_RELU[i] ? 1 : 0), S > 1 ? 0 : pk, himg)
#define SF_CONV(TM_, TN_)                                                                                           \
  do {                                                                                                              \
    if (m->split_f16) {                                                                                             \
      if (cfg.ck == 64) SF_CONV_H(TM_, TN_, 64); else SF_CONV_H(TM_, TN_, 32);                                       \
    } else                                                                                                          \
      hipLaunchKernelGGL((k_conv_igemm<9, TM_, TN_>), dim3((P + TM_ - 1) / TM_, (co + TN_ - 1) / TN_, S), dim3(256), 0, \
                         c->stream, src, h, w, VGG_CIN[i], (const float*)m->conv_w[i].p, co,                          \
                         S > 1 ? (const float*)nullptr : (const float*)m->conv_b[i].p, cdst,                          \
                         S > 1 ? 0 : (VGG_RELU[i] ? 1 : 0));                                                          \
  } while (0)
  if (tm == 128 && tn == 128) SF_CONV(128, 128);
  else if (tm == 128) SF_CONV(128, 64);
  else if (tn == 128) SF_CONV(64, 128);
  else SF_CONV(64, 64);
#undef SF_CONV
#undef SF_CONV_H
  if (S > 1) {
    const size_t n = (size_t)P * co;
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, c->stream,
                       (const float*)cdst, S, n, co, (const float*)m->conv_b[i].p, dst, VGG_RELU[i] ? 1 : 0, pk);
  }
  return SF_OK;
}

// Tap split of a layer: a FIXED function of its shape.  The split decides the order in which a pixel's taps are summed
// (S partial sums, then their sum), i.e. the low bits of the descriptor; it must not depend on a timing, or two robots
// -- or two handles of one -- would produce descriptors that differ in the last bits and near-tie NetVLAD matches could
// flip.  Rule: split the nine taps three ways when 128 x 128 tiles of the unsplit layer would not give every compute unit
// one workgroup (the 80 x 60 and 40 x 30 layers of a 640 x 480 image: with the split the 80 x 60 layers run wider tiles on
// more workgroups -- NetVLAD inference 1.49 -> 1.36 ms), and only while the partial outputs stay small.
// (Tried in round 3 and dropped: two LDS stages with ONE barrier per K step instead of one stage with two -- 1.41-1.44
//  against 1.36 ms on the same box: the second stage halves the workgroups a CU holds, and these kernels hide their
//  operand latency with occupancy.)
static int conv_split_rule(int P, int co) {
  const long wgs = (long)((P + 127) / 128) * ((co + 127) / 128);
  return (wgs < 256 && (size_t)P * co * 3 <= ((size_t)9 << 20)) ? 3 : 1;
}

// Tile of every layer for one image size, MEASURED: the first inference at a new size times each layer with tiles
// 64 / 128 x 64 / 128 and K steps of 32 / 64 channels (the tap split fixed by conv_split_rule: neither changes the order of
// a pixel's sum, so the choice is invisible in the results) on the buffers it is about to use and keeps the fastest (HIP events, three
// runs each; a few ms once per image size, synchronous).  The split-fp16 kernels are bound by the latency of their
// operand fetches, where more, smaller workgroups often win, and a measurement is the honest way to choose.
static int conv_autotune(sf_context* c, sf_netvlad_model* m, int H, int W, int n_stack, sf_netvlad_model::Tune& T) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) return sf_fail(c, SF_EHIP, "autotune: hipEventCreate failed");
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return sf_fail(c, SF_EHIP, "autotune: hipEventCreate failed"); }
  int h = H * n_stack, himg = H, w = W, rc = SF_OK;   // (a batch runs as a vertical stack of its images)
  const float* src = (const float*)m->act[0].p + CONV_ZERO_PAGE;
  float* dst = (float*)m->act[1].p + CONV_ZERO_PAGE;
  for (int i = 0; i < 13 && rc == SF_OK; ++i) {
    if (i > 0) {
      const int co = VGG_COUT[i];
      const int S = conv_split_rule(himg * w, co);      // (of ONE image's shape: a batch gives the bits of single calls)
      float best = 1e30f;
      ConvCfg pick = {64, 64, S, 32};
      for (int ck = 32; ck <= (m->split_f16 ? 64 : 32) && rc == SF_OK; ck *= 2)
      for (int tm = 64; tm <= 128 && rc == SF_OK; tm *= 2)
        for (int tn = 64; tn <= 128 && rc == SF_OK; tn *= 2) {
          if (co == 64 && tn == 128) continue;
          const ConvCfg cfg = {tm, tn, S, ck};
          rc = conv_layer(c, m, i, src, h, w, dst, cfg, himg);      // warm-up (and buffer growth)
          float t_min = 1e30f;
          for (int rep = 0; rep < 3 && rc == SF_OK; ++rep) {
            (void)hipEventRecord(e0, c->stream);
            rc = conv_layer(c, m, i, src, h, w, dst, cfg, himg);
            (void)hipEventRecord(e1, c->stream);
            if (rc == SF_OK && hipEventSynchronize(e1) != hipSuccess) rc = sf_fail(c, SF_EHIP, "autotune: event wait failed");
            float ms = 0.f;
            if (rc == SF_OK) (void)hipEventElapsedTime(&ms, e0, e1);
            t_min = std::min(t_min, ms);
          }
          if (rc == SF_OK && t_min < best) { best = t_min; pick = cfg; }
        }
      T.cfg[i] = pick;
    }
    if (VGG_POOL[i]) { h /= 2; himg /= 2; w /= 2; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc == SF_OK) { T.h = H; T.w = W; T.n = n_stack; }
  return rc;
}

// d_images: n images [H][W][3] float32 RGB on the device, back to back (what the reference feeds the placeholder,
// data_handler.py:60-61, netvlad_batch_size at a time: :149-156); d_out: n x n_out floats (the first n_out of each
// pca_dim-D unit vector).  Asynchronous on the handle's stream.  The trunk and the VLAD layer run image by image (every
// layer already fills the chip at camera resolution); the WPCA reads its weights ONCE per group of up to four images.
int sf_netvlad_infer_batch_impl(sf_context* c, const float* d_images, int n_img, int H, int W, float* d_out, int n_out) {
  sf_netvlad_model* m = c->netvlad;
  if (!m) return sf_fail(c, SF_EINVAL, "no NetVLAD model loaded (sf_netvlad_load)");
  if (n_img < 1 || n_img > 1024) return sf_fail(c, SF_ERANGE, "a batch of %d images", n_img);
  if (H < 16 || W < 16) return sf_fail(c, SF_ERANGE, "image of %d x %d is smaller than the four poolings need", W, H);
  if (n_out < 1 || n_out > m->pca_dim) return sf_fail(c, SF_ERANGE, "%d output dimensions of %d", n_out, m->pca_dim);
  int rc;
  const int K = m->clusters, D = 512;
  const int group = std::min(n_img, 4);
  // The trunk of a group runs as ONE vertical stack of its images (split-fp16 kernels; heights that survive the four
  // poolings as whole rows): the late layers of a camera image have too few pixels to fill the chip (40 x 30 at conv5),
  // a stack of three has three times the workgroups.  A tap never crosses from one image of the stack into the next
  // (the kernels take the image height), the tap split is the single image's, tiles and K steps do not enter the sums:
  // per image the results are the bits of the single-image call.
  // (k_conv_igemm_h addresses its input with 32-bit byte offsets: the largest activation tensor of a pass, 64 channels at
  //  full resolution, has to stay under 4 GB -- 16 Mpixel for a single image; a stack is cut down to what fits)
  const size_t img_bytes = (size_t)H * W * 64 * sizeof(float);
  const size_t off_limit = ((size_t)1 << 32) - 2 * CONV_ZERO_PAGE * sizeof(float);
  if (m->split_f16 && img_bytes > off_limit)
    return sf_fail(c, SF_ERANGE, "image of %d x %d is beyond the 32-bit activation offsets of the split-fp16 trunk", W, H);
  const bool stackable = m->split_f16 && (H % 16) == 0;
  const int ns_max = stackable ? (int)std::max<size_t>(1, std::min<size_t>((size_t)group, off_limit / img_bytes)) : 1;
  const size_t act_max = (size_t)ns_max * H * W * 64;
  // every activation buffer starts with a page of zeros: what k_conv_igemm_h reads for a tap outside the image
  for (int i = 0; i < 2; ++i) {
    if ((rc = sf_buf_reserve(c, m->act[i], (act_max + CONV_ZERO_PAGE) * sizeof(float))) != SF_OK) return rc;
    if (m->act[i].p != m->act_zeroed[i] || m->act[i].bytes != m->act_zeroed_bytes[i]) {   // (a new allocation)
      SF_HIP(c, hipMemsetAsync(m->act[i].p, 0, CONV_ZERO_PAGE * sizeof(float), c->stream));
      m->act_zeroed[i] = m->act[i].p;
      m->act_zeroed_bytes[i] = m->act[i].bytes;
    }
  }
  const int y_pitch = std::max(m->pca_dim, 64);
  if ((rc = sf_buf_reserve(c, m->vlad, (size_t)group * D * K * sizeof(float))) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, m->pca_y, (size_t)group * y_pitch * sizeof(float))) != SF_OK) return rc;
  for (int g0 = 0; g0 < n_img; g0 += group) {
    const int gb = std::min(group, n_img - g0);
    int ns = (stackable && gb > 1) ? std::min(gb, ns_max) : 1;          // images per pass of the trunk
    if (gb % ns != 0) ns = 1;                                          // (a stack cut down by the offset limit: whole passes only)
    sf_netvlad_model::Tune& T = m->tune[ns == 1 ? 0 : 1];
    if (T.h != H || T.w != W || T.n != ns) {
      if ((rc = conv_autotune(c, m, H, W, ns, T)) != SF_OK) return rc;
    }
    for (int b0 = 0; b0 < gb; b0 += ns) {
      const float* d_image = d_images + (size_t)(g0 + b0) * H * W * 3;
      int h = H * ns, himg = H, w = W, cur = 0;
      const float* src = d_image;
      for (int i = 0; i < 13; ++i) {
        float* dst = (float*)m->act[cur].p + CONV_ZERO_PAGE;
        const int P = h * w;
        if (i == 0) {
          hipLaunchKernelGGL(k_conv3x3_first, dim3((unsigned)(((size_t)P * 4 + 255) / 256)), dim3(256), 0, c->stream, src, h, w,
                             (const float*)m->conv_w[0].p, (const float*)m->conv_b[0].p, (const float*)m->mean.p, dst,
                             VGG_RELU[0] ? 1 : 0, m->split_f16 ? 1 : 0, himg);
        } else {
          if ((rc = conv_layer(c, m, i, src, h, w, dst, T.cfg[i], himg)) != SF_OK) return rc;
        }
        src = dst;
        cur ^= 1;
        if (VGG_POOL[i]) {
          float* pd = (float*)m->act[cur].p + CONV_ZERO_PAGE;
          const size_t n = (size_t)(h / 2) * (w / 2) * (VGG_COUT[i] / 4);
          hipLaunchKernelGGL(k_pool2_relu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, h, w, VGG_COUT[i], pd,
                             m->split_f16 ? 1 : 0);
          h /= 2; himg /= 2; w /= 2;
          src = pd;
          cur ^= 1;
        }
      }
      // src = conv5_3 output of the stack [ns][himg][w][512] (in act[cur ^ 1]); per image: normalise in place, assignment
      // into act[cur], aggregation
      const int P = himg * w;
      for (int j = 0; j < ns; ++j) {
        float* vlad = (float*)m->vlad.p + (size_t)(b0 + j) * D * K;
        float* x = const_cast<float*>(src) + (size_t)j * P * D;
        float* a = (float*)m->act[cur].p + CONV_ZERO_PAGE;
        float* norms = (float*)m->pca_y.p + (size_t)(b0 + j) * y_pitch;                     // (the slot doubles as the K norms)
        hipLaunchKernelGGL(k_l2norm_rows, dim3((P + 3) / 4), dim3(256), 0, c->stream, x, P, D);
        hipLaunchKernelGGL((k_conv_igemm<1, 64, 64>), dim3((P + 63) / 64, (K + 63) / 64), dim3(256), 0, c->stream, (const float*)x, himg, w, D,
                           (const float*)m->assign_w.p, K, (const float*)nullptr, a, 0);
        hipLaunchKernelGGL(k_softmax_rows, dim3((P + 3) / 4), dim3(256), 0, c->stream, a, P, K);
        hipLaunchKernelGGL(k_vlad_aggregate, dim3(D / 2), dim3(256), 0, c->stream, (const float*)x, (const float*)a,
                           (const float*)m->centers.p, P, D, K, vlad);
        hipLaunchKernelGGL(k_vlad_cluster_norms, dim3(K), dim3(256), 0, c->stream, (const float*)vlad, D, K, norms);
        hipLaunchKernelGGL(k_vlad_normalize, dim3(1), dim3(1024), 0, c->stream, vlad, D, K, (const float*)norms);
      }
    }
    const dim3 wg((m->pca_dim + 3) / 4);
    const float* V = (const float*)m->vlad.p;
    float* Y = (float*)m->pca_y.p;
    if (gb == 1)
      hipLaunchKernelGGL(k_wpca, wg, dim3(256), 0, c->stream, (const float*)m->pca_w.p, (const float*)m->pca_b.p, V, m->pca_dim, D * K, Y);
    else if (gb == 2)
      hipLaunchKernelGGL(k_wpca_batch<2>, wg, dim3(256), 0, c->stream, (const float*)m->pca_w.p, (const float*)m->pca_b.p, V, m->pca_dim, D * K, Y, y_pitch);
    else if (gb == 3)
      hipLaunchKernelGGL(k_wpca_batch<3>, wg, dim3(256), 0, c->stream, (const float*)m->pca_w.p, (const float*)m->pca_b.p, V, m->pca_dim, D * K, Y, y_pitch);
    else
      hipLaunchKernelGGL(k_wpca_batch<4>, wg, dim3(256), 0, c->stream, (const float*)m->pca_w.p, (const float*)m->pca_b.p, V, m->pca_dim, D * K, Y, y_pitch);
    for (int b = 0; b < gb; ++b)
      hipLaunchKernelGGL(k_l2norm_vec, dim3(1), dim3(256), 0, c->stream, (const float*)(Y + (size_t)b * y_pitch), m->pca_dim,
                         d_out + (size_t)(g0 + b) * n_out, n_out);
  }
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

int sf_netvlad_infer_impl(sf_context* c, const float* d_image, int H, int W, float* d_out, int n_out) {
  return sf_netvlad_infer_batch_impl(c, d_image, 1, H, W, d_out, n_out);
}
