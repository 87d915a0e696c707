// Where should the matcher's streamed operand come from?  (timing only; results are summed into a sink)
// The product's scan (k_match.hip knn2_mfma_tile) stages a pair's raw "from" descriptors (16 KB at K = 500) in LDS once
// and every wavefront spreads every 32-row tile into fp4 itself: 16 of the loop's ~63 vector instructions, four times
// redundantly per workgroup.  Variant G reads the tile ALREADY SPREAD (a 64 KB fp4 copy of the descriptors kept beside
// the raw ones in the keyframe store, tile-major, 1 KiB per (tile, k-step) in lane order) straight from global memory
// into the MFMA operand registers: no spread, no LDS, four times the bytes through L1 / L2.
// Both variants run the whole chip like the fused kernel does: 256-thread workgroups, 4 per CU (128 VGPRs), every
// workgroup walks its own sequence of pairs (fresh descriptor blocks each), 2 column groups x 16 tiles per wavefront.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/match_stream.hip -o tools/ubench/match_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v8i spread_from(unsigned x, unsigned m88, unsigned c22) {
  v8i o = {0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (int)((x & m88) | c22); o[1] = (int)(x & 0x44444444u); o[2] = (int)(x & 0x22222222u); o[3] = (int)(x & 0x11111111u);
  return o;
}
__device__ __forceinline__ v8i spread_to(unsigned y, unsigned m88, unsigned c22) {
  v8i o = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned n = ~y;
  o[0] = (int)((y & m88) | c22); o[1] = (int)(((n << 1) & m88) | c22);
  o[2] = (int)(((n << 2) & m88) | 0x44444444u); o[3] = (int)(((n << 3) & m88) | 0x66666666u);
  return o;
}
__device__ __forceinline__ void top2_16(const v16f& v, float& b, float& s) {
  const float x0 = __builtin_amdgcn_fmed3f(b, v[0], v[1]);
  float ta, tb;
  asm("v_max3_f32 %0, %0, %4, %5\n\tv_med3_f32 %2, %0, %6, %7\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %20, %2\n\t"
      "v_med3_f32 %2, %0, %8, %9\n\tv_max3_f32 %0, %0, %8, %9\n\tv_med3_f32 %3, %0, %10, %11\n\tv_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %1, %1, %2, %3\n\tv_med3_f32 %2, %0, %12, %13\n\tv_max3_f32 %0, %0, %12, %13\n\tv_med3_f32 %3, %0, %14, %15\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\tv_max3_f32 %1, %1, %2, %3\n\tv_med3_f32 %2, %0, %16, %17\n\tv_max3_f32 %0, %0, %16, %17\n\t"
      "v_med3_f32 %3, %0, %18, %19\n\tv_max3_f32 %0, %0, %18, %19\n\tv_max3_f32 %1, %1, %2, %3"
      : "+v"(b), "+v"(s), "=&v"(ta), "=&v"(tb)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
        "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(x0));
}

constexpr int KF = 512;                 // rows per descriptor block (16 tiles of 32)
constexpr int RAW_DW = KF * 8;          // dwords of a raw block (256-bit rows)
constexpr int FP4_DW = KF * 32;         // dwords of a spread block

// S: the product's form.  raw: [blocks][KF][8] dwords
__global__ void __launch_bounds__(256, 4) kS(const unsigned* __restrict__ raw, int n_blocks, int pairs_per_wg, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned lds[RAW_DW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  float total = 0.f;
  for (int p = 0; p < pairs_per_wg; ++p) {
    const int bf = (blockIdx.x + p * gridDim.x) % n_blocks, bt = (bf + 7919) % n_blocks;
    const unsigned* dF = raw + (size_t)bf * RAW_DW;
    const unsigned* dT = raw + (size_t)bt * RAW_DW;
    __syncthreads();
    for (int i = tid; i < RAW_DW / 4; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(dF)[i];
    __syncthreads();
    for (int g = 0; g < 2; ++g) {
      v8i Bf[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = (wave + 4 * (2 * g + j)) * 32 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(dT + (size_t)t * 8 + 4 * h);
        Bf[j][0] = spread_to(v.x, m88, c22); Bf[j][1] = spread_to(v.y, m88, c22);
        Bf[j][2] = spread_to(v.z, m88, c22); Bf[j][3] = spread_to(v.w, m88, c22);
      }
      float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
#pragma unroll
      for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
      uint4 rw = *reinterpret_cast<const uint4*>(lds + r * 8 + 4 * h);
#pragma unroll 1
      for (int mt = 0; mt < 16; ++mt) {
        v16f c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = cin[i];
        v8i Af[4];
        Af[0] = spread_from(rw.x, m88, c22); Af[1] = spread_from(rw.y, m88, c22);
        Af[2] = spread_from(rw.z, m88, c22); Af[3] = spread_from(rw.w, m88, c22);
        rw = *reinterpret_cast<const uint4*>(lds + (min(mt + 1, 15) * 32 + r) * 8 + 4 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[j][0], c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
          for (int k = 1; k < 4; ++k) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[k], Bf[j][k], acc, 4, 4, 0, 0, 0, 0);
          top2_16(acc, b[j], s[j]);
        }
      }
      total += (b[0] + s[0]) + (b[1] + s[1]);
    }
  }
  sink[blockIdx.x * 256 + tid] = total;
}

// R: raw rows straight from global memory (L2), one tile ahead, no LDS staging and no barrier: every wavefront reads
// the pair's 16 KB itself (1 KiB, perfectly coalesced, per tile).  raw: [blocks][KF][8] dwords
__global__ void __launch_bounds__(256, 4) kR(const unsigned* __restrict__ raw, int n_blocks, int pairs_per_wg, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  float total = 0.f;
  for (int p = 0; p < pairs_per_wg; ++p) {
    const int bf = (blockIdx.x + p * gridDim.x) % n_blocks, bt = (bf + 7919) % n_blocks;
    const unsigned* dF = raw + (size_t)bf * RAW_DW;
    const unsigned* dT = raw + (size_t)bt * RAW_DW;
    for (int g = 0; g < 2; ++g) {
      v8i Bf[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = (wave + 4 * (2 * g + j)) * 32 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(dT + (size_t)t * 8 + 4 * h);
        Bf[j][0] = spread_to(v.x, m88, c22); Bf[j][1] = spread_to(v.y, m88, c22);
        Bf[j][2] = spread_to(v.z, m88, c22); Bf[j][3] = spread_to(v.w, m88, c22);
      }
      float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
#pragma unroll
      for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
      uint4 rw = *reinterpret_cast<const uint4*>(dF + r * 8 + 4 * h);
#pragma unroll 1
      for (int mt = 0; mt < 16; ++mt) {
        v16f c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = cin[i];
        v8i Af[4];
        Af[0] = spread_from(rw.x, m88, c22); Af[1] = spread_from(rw.y, m88, c22);
        Af[2] = spread_from(rw.z, m88, c22); Af[3] = spread_from(rw.w, m88, c22);
        rw = *reinterpret_cast<const uint4*>(dF + (min(mt + 1, 15) * 32 + r) * 8 + 4 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[j][0], c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
          for (int k = 1; k < 4; ++k) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[k], Bf[j][k], acc, 4, 4, 0, 0, 0, 0);
          top2_16(acc, b[j], s[j]);
        }
      }
      total += (b[0] + s[0]) + (b[1] + s[1]);
    }
  }
  sink[blockIdx.x * 256 + tid] = total;
}

// G: the "from" operand already spread, from global memory.  fp4: [blocks][16 tiles][4 k-steps][64 lanes][4 dwords]
template <int AHEAD>   // 0: a tile's operands are requested when the previous tile's MFMAs have consumed the registers
                       // (no extra registers); 1: one tile ahead in a second register set (+16 VGPRs)
__global__ void __launch_bounds__(256, 4) kG(const unsigned* __restrict__ raw, const uint4* __restrict__ fp4, int n_blocks,
                                             int pairs_per_wg, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  float total = 0.f;
  for (int p = 0; p < pairs_per_wg; ++p) {
    const int bf = (blockIdx.x + p * gridDim.x) % n_blocks, bt = (bf + 7919) % n_blocks;
    const uint4* dF = fp4 + (size_t)bf * (FP4_DW / 4) + lane;
    const unsigned* dT = raw + (size_t)bt * RAW_DW;
    for (int g = 0; g < 2; ++g) {
      v8i Bf[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = (wave + 4 * (2 * g + j)) * 32 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(dT + (size_t)t * 8 + 4 * h);
        Bf[j][0] = spread_to(v.x, m88, c22); Bf[j][1] = spread_to(v.y, m88, c22);
        Bf[j][2] = spread_to(v.z, m88, c22); Bf[j][3] = spread_to(v.w, m88, c22);
      }
      float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
#pragma unroll
      for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
      uint4 a0 = dF[0], a1 = dF[64], a2 = dF[128], a3 = dF[192];
#pragma unroll 1
      for (int mt = 0; mt < 16; ++mt) {
        v16f c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = cin[i];
        const v8i A0 = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
        const v8i A1 = {(int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w, 0, 0, 0, 0};
        const v8i A2 = {(int)a2.x, (int)a2.y, (int)a2.z, (int)a2.w, 0, 0, 0, 0};
        const v8i A3 = {(int)a3.x, (int)a3.y, (int)a3.z, (int)a3.w, 0, 0, 0, 0};
        const uint4* nx = dF + (size_t)min(mt + 1, 15) * 256;
        if (AHEAD) { a0 = nx[0]; a1 = nx[64]; a2 = nx[128]; a3 = nx[192]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
        v16f acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, Bf[0][0], c0, 4, 4, 0, 0, 0, 0);
        v16f acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, Bf[1][0], c0, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, Bf[0][1], acc0, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, Bf[1][1], acc1, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, Bf[0][2], acc0, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, Bf[1][2], acc1, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A3, Bf[0][3], acc0, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A3, Bf[1][3], acc1, 4, 4, 0, 0, 0, 0);
        if (!AHEAD) {
          __builtin_amdgcn_sched_barrier(0);          // the requests stay BEHIND the MFMAs that read the registers
          a0 = nx[0]; a1 = nx[64]; a2 = nx[128]; a3 = nx[192];
          __builtin_amdgcn_sched_barrier(0);
        }
        top2_16(acc0, b[0], s[0]);
        top2_16(acc1, b[1], s[1]);
      }
      total += (b[0] + s[0]) + (b[1] + s[1]);
    }
  }
  sink[blockIdx.x * 256 + tid] = total;
}


// L: the spread shared by the workgroup through LDS.  Wavefront w spreads k-step w of every 32-row tile (one raw dword
// per lane, straight from global memory: 4 vector instructions instead of 16) into a two-tile ring (8 KB; the 16 KB raw
// staging block is gone); all four wavefronts read the whole tile back (4 ds_read_b128, in place, behind the MFMAs that
// consumed the previous one).  One workgroup barrier per tile.
template <int BAR>   // 1: the product-correct form; 0: no per-tile barrier (timing only: what the barrier costs)
__global__ void __launch_bounds__(256, 4) kL(const unsigned* __restrict__ raw, int n_blocks, int pairs_per_wg, float* sink) {
  __shared__ uint4 ring[2][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  float total = 0.f;
  auto spread4 = [&](unsigned x) {
    uint4 o;
    o.x = (x & m88) | c22; o.y = x & 0x44444444u; o.z = x & 0x22222222u; o.w = x & 0x11111111u;
    return o;
  };
  for (int p = 0; p < pairs_per_wg; ++p) {
    const int bf = (blockIdx.x + p * gridDim.x) % n_blocks, bt = (bf + 7919) % n_blocks;
    const unsigned* src = raw + (size_t)bf * RAW_DW + r * 8 + 4 * h + wave;      // + 256 per tile
    const unsigned* dT = raw + (size_t)bt * RAW_DW;
    for (int g = 0; g < 2; ++g) {
      v8i Bf[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = (wave + 4 * (2 * g + j)) * 32 + r;
        const uint4 v = *reinterpret_cast<const uint4*>(dT + (size_t)t * 8 + 4 * h);
        Bf[j][0] = spread_to(v.x, m88, c22); Bf[j][1] = spread_to(v.y, m88, c22);
        Bf[j][2] = spread_to(v.z, m88, c22); Bf[j][3] = spread_to(v.w, m88, c22);
      }
      float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
#pragma unroll
      for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
      unsigned rw0 = src[0], rw1 = src[256], rw2 = src[512], rw3 = src[768];
      __syncthreads();                                  // the ring is free (the previous group's reads have landed)
      ring[0][wave][lane] = spread4(rw0);
      ring[1][wave][lane] = spread4(rw1);
      __syncthreads();
      uint4 a0 = ring[0][0][lane], a1 = ring[0][1][lane], a2 = ring[0][2][lane], a3 = ring[0][3][lane];
#pragma unroll 1
      for (int mt = 0; mt < 16; ++mt) {
        v16f c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c0[i] = cin[i];
        const v8i A0 = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, 0, 0, 0, 0};
        const v8i A1 = {(int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w, 0, 0, 0, 0};
        const v8i A2 = {(int)a2.x, (int)a2.y, (int)a2.z, (int)a2.w, 0, 0, 0, 0};
        const v8i A3 = {(int)a3.x, (int)a3.y, (int)a3.z, (int)a3.w, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
        v16f acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, Bf[0][0], c0, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, Bf[0][1], acc0, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, Bf[0][2], acc0, 4, 4, 0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A3, Bf[0][3], acc0, 4, 4, 0, 0, 0, 0);
        v16f acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, Bf[1][0], c0, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, Bf[1][1], acc1, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A2, Bf[1][2], acc1, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A3, Bf[1][3], acc1, 4, 4, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);            // the requests stay BEHIND the MFMAs that read the registers
        const int sn = (mt + 1) & 1;                   // tile mt + 1: written one iteration ago, behind a barrier
        a0 = ring[sn][0][lane]; a1 = ring[sn][1][lane]; a2 = ring[sn][2][lane]; a3 = ring[sn][3][lane];
        __builtin_amdgcn_sched_barrier(0);
        top2_16(acc0, b[0], s[0]);
        // tile mt + 2 replaces tile mt (every wavefront's reads of it landed before the last barrier)
        ring[mt & 1][wave][lane] = spread4(rw2);
        rw2 = rw3;
        rw3 = src[(size_t)min(mt + 4, 15) * 256];
        top2_16(acc1, b[1], s[1]);
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      total += (b[0] + s[0]) + (b[1] + s[1]);
    }
  }
  sink[blockIdx.x * 256 + tid] = total;
}

int main(int argc, char** argv) {
  const int n_blocks = argc > 1 ? atoi(argv[1]) : 8192;     // 8192 x (16 + 64) KB = 640 MB: past the Infinity Cache
  const int ppw = argc > 2 ? atoi(argv[2]) : 16;
  unsigned* raw; uint4* fp4; float* sink;
  (void)hipMalloc(&raw, (size_t)n_blocks * RAW_DW * 4);
  (void)hipMalloc(&fp4, (size_t)n_blocks * FP4_DW * 4);
  (void)hipMalloc(&sink, 4096 * 256 * 4);
  {
    std::vector<unsigned> hbuf((size_t)n_blocks * RAW_DW);
    unsigned x = 12345u;
    for (auto& v : hbuf) { x = x * 1664525u + 1013904223u; v = x; }
    (void)hipMemcpy(raw, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice);
    // (the fp4 copy's CONTENT does not matter for the timing; any finite fp4 pattern will do)
    std::vector<unsigned> h4((size_t)n_blocks * FP4_DW);
    for (auto& v : h4) { x = x * 1664525u + 1013904223u; v = x & 0xBBBBBBBBu; }
    (void)hipMemcpy(fp4, h4.data(), h4.size() * 4, hipMemcpyHostToDevice);
  }
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double pairs = (double)grid * ppw;
    printf("%-46s %8.3f ms for %6.0f pairs = %6.2f M pairs/s (matching scan only), %5.1f us of a CU per pair\n", name, best,
           pairs, pairs / best / 1e3, best * 1e3 * cus / pairs);
  };
  timeit("S  raw rows in LDS, spread per wavefront", [&] { hipLaunchKernelGGL(kS, dim3(grid), dim3(256), 0, 0, raw, n_blocks, ppw, sink); });
  timeit("R  raw rows from global per wavefront, no LDS", [&] { hipLaunchKernelGGL(kR, dim3(grid), dim3(256), 0, 0, raw, n_blocks, ppw, sink); });
  timeit("G0 spread rows from global, in-place reload", [&] { hipLaunchKernelGGL(kG<0>, dim3(grid), dim3(256), 0, 0, raw, fp4, n_blocks, ppw, sink); });
  timeit("G1 spread rows from global, one tile ahead", [&] { hipLaunchKernelGGL(kG<1>, dim3(grid), dim3(256), 0, 0, raw, fp4, n_blocks, ppw, sink); });
  timeit("L  spread shared through a two-tile LDS ring", [&] { hipLaunchKernelGGL(kL<1>, dim3(grid), dim3(256), 0, 0, raw, n_blocks, ppw, sink); });
  timeit("L' the same without its per-tile barrier", [&] { hipLaunchKernelGGL(kL<0>, dim3(grid), dim3(256), 0, 0, raw, n_blocks, ppw, sink); });
  return 0;
}
