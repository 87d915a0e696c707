// The matcher's inner loop in isolation (timing only, the results are not checked here): one "from" tile out of LDS
// against resident "to" columns, exact +-1 products on the fp4 matrix cores, top-2 merge of every accumulator.
//   A  the product's form (k_match.hip knn2_mfma_tile): 32 from rows x 64 to columns per iteration,
//      v_mfma_f32_32x32x64_f8f6f4, 4 chained MFMAs per 32 x 32 block, the 20-op merge behind each chain;
//   B  16 from rows x 64 to columns per iteration on v_mfma_f32_16x16x128_f8f6f4: four 16 x 16 blocks with a 2-MFMA
//      chain each, issued round-robin, so that the merge of a block (5 ops on its 4 accumulator registers) runs while
//      the other blocks' MFMAs execute; the next tile's raw rows are requested one iteration ahead;
//   C  B with 128 resident columns (8 blocks).
// Cycles per iteration from s_memtime for 1 .. 4 wavefronts per SIMD (one workgroup on one CU), plus the implied
// share of the matrix pipe: A does 8 MFMAs of 32 cycles per iteration, B 8 of 16, C 16 of 16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/match_loop.hip -o tools/ubench/match_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v8i spread_from(unsigned x, unsigned m88, unsigned c22) {
  v8i o = {0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (int)((x & m88) | c22); o[1] = (int)(x & 0x44444444u); o[2] = (int)(x & 0x22222222u); o[3] = (int)(x & 0x11111111u);
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
// (b, s) <- the two largest of {b, s, v[0..3]}: 5 ops; the first read of the accumulator is a builtin (hazard states)
__device__ __forceinline__ void top2_4(const v4f& v, float& b, float& s) {
  const float x0 = __builtin_amdgcn_fmed3f(b, v[0], v[1]);
  float t;
  asm("v_max3_f32 %0, %0, %3, %4\n\tv_med3_f32 %2, %0, %5, %6\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %7, %2"
      : "+v"(b), "+v"(s), "=&v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(x0));
}

__global__ void __launch_bounds__(1024) kA(unsigned long long* out, float* sink, const unsigned* g, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  v8i Bf[2][4];
  for (int j = 0; j < 2; ++j) for (int k = 0; k < 4; ++k) Bf[j][k] = spread_from(g[lane * 8 + j * 4 + k] ^ 0x5a5a5a5au, m88, c22);
  float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
  for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int mt = it & 15;
    const uint4 rw = *reinterpret_cast<const uint4*>(lds + (mt * 32 + r) * 8 + 4 * h);
    const unsigned raw[4] = {rw.x, rw.y, rw.z, rw.w};
    v16f c0;
    for (int i = 0; i < 16; ++i) c0[i] = cin[i];
    v8i Af[4];
    for (int k = 0; k < 4; ++k) Af[k] = spread_from(raw[k], m88, c22);
    for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[j][0], c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
      for (int k = 1; k < 4; ++k) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[k], Bf[j][k], acc, 4, 4, 0, 0, 0, 0);
      top2_16(acc, b[j], s[j]);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = b[0] + s[0] + b[1] + s[1];
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// (b, s) <- the two largest of {b, s, v[i0..i0+3]}: one quarter of top2_16, so that the merge of a finished chain can be
// spread between the MFMAs of the next chain (variant D)
__device__ __forceinline__ void top2_q(float v0, float v1, float v2, float v3, float& b, float& s, bool first) {
  float x0 = first ? __builtin_amdgcn_fmed3f(b, v0, v1) : 0.f;
  float t;
  if (first)
    asm("v_max3_f32 %0, %0, %3, %4\n\tv_med3_f32 %2, %0, %5, %6\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %7, %2"
        : "+v"(b), "+v"(s), "=&v"(t) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(x0));
  else
    asm("v_med3_f32 %2, %0, %3, %4\n\tv_max3_f32 %0, %0, %3, %4\n\tv_med3_f32 %7, %0, %5, %6\n\tv_max3_f32 %0, %0, %5, %6\n\t"
        "v_max3_f32 %1, %1, %2, %7"
        : "+v"(b), "+v"(s), "=&v"(t), "=&v"(x0) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
}

// D: A's arithmetic with the instruction order chosen by hand: chain 0's four MFMAs first, then chain 1's with the merge
// of chain 0 spread between them (a quarter behind each), the next tile's rows requested and spread behind chain 1's last
// MFMA, the merge of chain 1 last -- a wavefront then feeds the matrix pipe for all but ~36 VALU instructions per tile.
__global__ void __launch_bounds__(1024) kD(unsigned long long* out, float* sink, const unsigned* g, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  v8i Bf[2][4];
  for (int j = 0; j < 2; ++j) for (int k = 0; k < 4; ++k) Bf[j][k] = spread_from(g[lane * 8 + j * 4 + k] ^ 0x5a5a5a5au, m88, c22);
  float cin[16], b[2] = {-1e30f, -1e30f}, s[2] = {-1e30f, -1e30f};
  for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) / 2048.f;
  uint4 rw = *reinterpret_cast<const uint4*>(lds + r * 8 + 4 * h);
  v8i Af[4];
  Af[0] = spread_from(rw.x, m88, c22); Af[1] = spread_from(rw.y, m88, c22);
  Af[2] = spread_from(rw.z, m88, c22); Af[3] = spread_from(rw.w, m88, c22);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int mt = (it + 1) & 15;
    v16f c0;
    for (int i = 0; i < 16; ++i) c0[i] = cin[i];
    for (int j = 0; j < 2; ++j) { b[j] += 32.f / 2048.f; s[j] += 32.f / 2048.f; }
    v16f a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[0][0], c0, 4, 4, 0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[1], Bf[0][1], a0, 4, 4, 0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[2], Bf[0][2], a0, 4, 4, 0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[3], Bf[0][3], a0, 4, 4, 0, 0, 0, 0);
    v16f a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[1][0], c0, 4, 4, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    top2_q(a0[0], a0[1], a0[2], a0[3], b[0], s[0], true);
    __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[1], Bf[1][1], a1, 4, 4, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    top2_q(a0[4], a0[5], a0[6], a0[7], b[0], s[0], false);
    __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[2], Bf[1][2], a1, 4, 4, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    top2_q(a0[8], a0[9], a0[10], a0[11], b[0], s[0], false);
    __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[3], Bf[1][3], a1, 4, 4, 0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    top2_q(a0[12], a0[13], a0[14], a0[15], b[0], s[0], false);
    // the next tile: rows out of LDS and their spread, while chain 1 finishes
    rw = *reinterpret_cast<const uint4*>(lds + (mt * 32 + r) * 8 + 4 * h);
    Af[0] = spread_from(rw.x, m88, c22); Af[1] = spread_from(rw.y, m88, c22);
    Af[2] = spread_from(rw.z, m88, c22); Af[3] = spread_from(rw.w, m88, c22);
    __builtin_amdgcn_sched_barrier(0);
    top2_16(a1, b[1], s[1]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[threadIdx.x] = b[0] + s[0] + b[1] + s[1] + (float)Af[0][0];
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// NB resident 16-column blocks per wavefront (4: 64 columns, 8: 128 columns)
template <int NB>
__global__ void __launch_bounds__(1024) kB(unsigned long long* out, float* sink, const unsigned* g, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  unsigned m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  v8i Bf[NB][2];
  for (int j = 0; j < NB; ++j) for (int k = 0; k < 2; ++k) Bf[j][k] = spread_from(g[lane * 16 + j * 2 + k] ^ 0x5a5a5a5au, m88, c22);
  v4f cin;
  for (int i = 0; i < 4; ++i) cin[i] = -(float)(4 * q + i) / 2048.f;
  float b[NB], s[NB];
  for (int j = 0; j < NB; ++j) { b[j] = -1e30f; s[j] = -1e30f; }
  // row r of the tile, dwords q and 4 + q of its 8 (k-step 0: bits 0..127 = dwords 0..3, k-step 1: dwords 4..7)
  uint2 nxt = make_uint2(lds[r * 8 + q], lds[r * 8 + 4 + q]);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint2 raw = nxt;
    const int mt = (it + 1) & 31;
    nxt = make_uint2(lds[(mt * 16 + r) * 8 + q], lds[(mt * 16 + r) * 8 + 4 + q]);     // next tile, one iteration ahead
    const v8i A0 = spread_from(raw.x, m88, c22), A1 = spread_from(raw.y, m88, c22);
    v4f acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A0, Bf[j][0], cin, 4, 4, 0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A1, Bf[j][1], acc[j], 4, 4, 0, 0, 0, 0);
      b[j] += 16.f / 2048.f; s[j] += 16.f / 2048.f;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) top2_4(acc[j], b[j], s[j]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float t = 0.f;
  for (int j = 0; j < NB; ++j) t += b[j] + s[j];
  sink[threadIdx.x] = t + (float)nxt.x;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

// calibration: 8 MFMAs per iteration in two chains, nothing else
__global__ void __launch_bounds__(1024) kM(unsigned long long* out, float* sink, const unsigned* g, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned m88 = 0x88888888u, c22 = 0x22222222u;
  v8i A = spread_from(g[lane], m88, c22), B = spread_from(g[lane + 64], m88, c22);
  v16f a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 1.f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, a0, 4, 4, 0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, a1, 4, 4, 0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += a0[i] + a1[i];
  sink[threadIdx.x] = t;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int waves_per_simd, double mfma_cycles_per_iter, double cells_per_iter, const unsigned* g) {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 8); (void)hipMalloc(&sink, 4096 * 4);
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves_per_simd), 16384, 0, d, sink, g, 2000);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(1), dim3(256 * waves_per_simd), 16384, 0, d, sink, g, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double cyc = (double)h / iters, ns = ms * 1e6 / iters;
  // (peak of the fp4 pipe: 2048 multiply-adds per cycle and SIMD at 2.4 GHz = 8 cells of 256 bit products per cycle)
  printf("%-28s waves/SIMD %d : %7.1f ticks, %7.1f ns per iteration per wave; %5.2f cells per ns and SIMD (peak 19.2); "
         "matrix pipe %3.0f %% busy at 2.4 GHz\n",
         name, waves_per_simd, cyc, ns, waves_per_simd * cells_per_iter / ns,
         100.0 * waves_per_simd * mfma_cycles_per_iter / (ns * 2.4));
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  unsigned* g; (void)hipMalloc(&g, 65536 * 4);
  unsigned* hbuf = (unsigned*)malloc(65536 * 4);
  unsigned x = 12345u;
  for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; hbuf[i] = x; }
  (void)hipMemcpy(g, hbuf, 65536 * 4, hipMemcpyHostToDevice);
  for (int w = 1; w <= 4; ++w) run("M 8 MFMAs only", kM, w, 8 * 32.0, 32.0 * 64, g);
  for (int w = 1; w <= 4; ++w) run("A 32x32x64, 32 x 64 cells", kA, w, 8 * 32.0, 32.0 * 64, g);
  for (int w = 1; w <= 4; ++w) run("D = A, hand-ordered", kD, w, 8 * 32.0, 32.0 * 64, g);
  for (int w = 1; w <= 4; ++w) run("B 16x16x128, 16 x 64 cells", kB<4>, w, 8 * 16.0, 16.0 * 64, g);
  for (int w = 1; w <= 4; ++w) run("C 16x16x128, 16 x 128 cells", kB<8>, w, 16 * 16.0, 16.0 * 128, g);
  return 0;
}
