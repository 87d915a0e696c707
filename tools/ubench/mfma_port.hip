// How long does an fp4 MFMA (v_mfma_f32_32x32x64_f8f6f4, 8 passes = 32 cycles of the matrix pipe) hold the SIMD's vector
// issue port?  A loop of 8 MFMAs (two accumulator tuples alternating, four-deep chains from a fresh C operand like the
// matcher's scan) and N plain, independent vector instructions (v_and_b32 on registers of their own), run by 1 .. 4
// wavefronts per SIMD on every SIMD of the chip.  If the port is held 4 cycles by a vector instruction and c by an MFMA,
// an iteration takes max(256, 4 N + 8 c) cycles once enough wavefronts are resident: flat up to N* = 64 - 2 c, then 4
// cycles per instruction.  Timed with s_memtime (100 MHz) and reported in cycles of the clock the chip held.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_port.hip -o tools/ubench/mfma_port
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int N, bool MFMA = true>
__global__ void __launch_bounds__(256) k(int iters, float* sink, unsigned seed) {
  const int lane = threadIdx.x & 63;
  v8i A = {(int)(0x22222222u ^ (lane * 0x88888888u)), 0x2A2A2A2A, 0x22AA22AA, (int)0xA2A2A2A2u, 0, 0, 0, 0};
  v8i B = {(int)(0x2222AAAAu + seed), 0x22222222, (int)0xAAAA2222u, 0x2A2A2A2A, 0, 0, 0, 0};
  v8i B1 = {(int)(0xA2A22222u + seed), 0x2222A2A2, (int)0xAA22AA22u, 0x22AA2A2A, 0, 0, 0, 0};   // (a second chain of its own)
  v16f c0;
#pragma unroll
  for (int i = 0; i < 16; ++i) c0[i] = (float)i;
  unsigned x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed * (i + 3) + lane;
  float total = 0.f;
  for (int it = 0; it < iters; ++it) {
    v16f a0 = c0, a1 = c0;
    if (MFMA) {
      a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c0, 4, 4, 0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B1, c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, a0, 4, 4, 0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B1, a1, 4, 4, 0, 0, 0, 0);
      }
    }
    // N plain vector instructions, none of which touches a register of the MFMAs
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 1) & 7]));
    // (one read of each accumulator per iteration keeps the chains alive; the matcher reads all of them)
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(total) : "v"(a0[0]));
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(total) : "v"(a1[0]));
  }
  unsigned y = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) y ^= x[i];
  sink[blockIdx.x * 256 + threadIdx.x] = total + (float)y;
}

template <int N>
void run(int cus, float* sink, double mhz) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  printf("N = %3d vector instructions + 8 MFMAs:", N);
  for (int wps = 1; wps <= 4; ++wps) {                 // 256-thread workgroups: one wavefront per SIMD each
    const int grid = cus * wps;
    hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, 200, sink, 7u);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, iters, sink, 7u);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    // cycles of one SIMD per iteration of ONE wavefront's loop, all resident wavefronts sharing it
    const double cyc = best * 1e-3 * mhz * 1e6 / iters / wps;
    printf("  %d/SIMD %6.1f", wps, cyc);
  }
  printf("   cycles per iteration and SIMD (at %.0f MHz)\n", mhz);
}

int main(int argc, char** argv) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* sink;
  (void)hipMalloc(&sink, (size_t)cus * 4 * 256 * 4);
  // the clock the chip holds: 256 plain vector instructions per iteration, no MFMA, 4 wavefronts per SIMD = 4 cycles each
  double mhz = 2400.0;
  {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<256, false>), dim3(cus * 4), dim3(256), 0, 0, 400, sink, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<256, false>), dim3(cus * 4), dim3(256), 0, 0, 4000, sink, 7u);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 4 wavefronts x 4000 iterations x 258 instructions x 4 cycles on every SIMD
    mhz = 4.0 * 4000 * 258 * 4 / (ms * 1e-3) / 1e6;
    printf("calibration: 258 plain vector instructions per iteration, 4 wavefronts per SIMD: %.3f ms -> %.0f MHz if each takes 4 cycles\n", ms, mhz);
  }
  run<0>(cus, sink, mhz); run<8>(cus, sink, mhz); run<16>(cus, sink, mhz); run<24>(cus, sink, mhz); run<28>(cus, sink, mhz);
  run<32>(cus, sink, mhz); run<36>(cus, sink, mhz); run<40>(cus, sink, mhz); run<44>(cus, sink, mhz); run<48>(cus, sink, mhz);
  run<56>(cus, sink, mhz); run<64>(cus, sink, mhz); run<80>(cus, sink, mhz); run<96>(cus, sink, mhz);
  return 0;
}
