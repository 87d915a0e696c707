// Issue interval of v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4) on one wavefront: a chain that accumulates into ONE
// register set (every MFMA waits for the one before), 2 and 4 interleaved chains, and the same with VALU work
// between the MFMAs.  Cycles from s_memtime, per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_fp4_latency.hip -o tools/ubench/mfma_fp4_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS, int VALU>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0};
  v8i b = {0x2a2a2222, 0x22222a2a, 0x2a2a2222, 0x2a2a2a2a, 0, 0, 0, 0};
  v16f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float x = threadIdx.x, y = 1.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = x;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  sink[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// MODE 1: the VALU work between the MFMAs is the matcher's top-2 merge of ANOTHER chain's accumulator (3-source
// v_med3 / v_max3 reading 4 accumulator registers per 5 instructions); MODE 2: the same instruction mix on plain
// registers (no accumulator reads).
template <int MODE>
__global__ void k2(unsigned long long* out, float* sink, int iters) {
  v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0};
  v8i b = {0x2a2a2222, 0x22222a2a, 0x2a2a2222, 0x2a2a2a2a, 0, 0, 0, 0};
  v16f accP, accQ, cin;
  for (int i = 0; i < 16; ++i) { accP[i] = 0.f; accQ[i] = 0.f; cin[i] = -(float)i; }
  float bb = -1e30f, ss = -1e30f, r0 = threadIdx.x, r1 = 2.f, r2 = 3.f, r3 = 4.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      v16f& acc = half ? accQ : accP;
      const v16f& old = half ? accP : accQ;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, kk == 0 ? cin : acc, 4, 4, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        float x, y;
        if (MODE == 1)
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_max3_f32 %0, %0, %4, %5\n\tv_med3_f32 %3, %0, %6, %7\n\t"
                       "v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %2, %3"
                       : "+v"(bb), "+v"(ss), "=&v"(x), "=&v"(y)
                       : "v"(old[4 * kk]), "v"(old[4 * kk + 1]), "v"(old[4 * kk + 2]), "v"(old[4 * kk + 3]));
        else
          asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_max3_f32 %0, %0, %4, %5\n\tv_med3_f32 %3, %0, %6, %7\n\t"
                       "v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %2, %3"
                       : "+v"(bb), "+v"(ss), "=&v"(x), "=&v"(y) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = bb + ss;
  for (int i = 0; i < 16; ++i) s += accP[i] + accQ[i];
  sink[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE>
void run2(int waves_per_simd) {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 8); (void)hipMalloc(&sink, 4 * 256 * 1024);
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<MODE>), dim3(1), dim3(256 * waves_per_simd), 0, 0, d, sink, 2000);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k2<MODE>), dim3(1), dim3(256 * waves_per_simd), 0, 0, d, sink, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("dependent chains of 4 + merge of the other accumulator (mode %d: %s)  waves/SIMD %d : %.1f ticks per MFMA per wave; "
         "%.1f ns per MFMA per wave (events), %.0f ticks/us\n",
         MODE, MODE == 1 ? "reads accumulators" : "plain registers", waves_per_simd, (double)h / (iters * 8.0),
         ms * 1e6 / (iters * 8.0), (double)h / (ms * 1e3));
  (void)hipFree(d); (void)hipFree(sink);
}

template <int CHAINS, int VALU>
void run(int waves_per_simd) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 8); hipMalloc(&sink, 4 * 256 * 1024);
  const int iters = 2000;
  // one workgroup of 64 * 4 * waves threads on one CU: waves_per_simd wavefronts on every SIMD
  hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(1), dim3(256 * waves_per_simd), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("chains %d  valu/mfma %d  waves/SIMD %d : %.1f cycles per MFMA per wave (shader cycles)\n",
         CHAINS, VALU, waves_per_simd, (double)h / (iters * CHAINS));
  hipFree(d); hipFree(sink);
}
int main() {
  run<1, 0>(1); run<2, 0>(1); run<4, 0>(1);
  run<1, 5>(1); run<2, 5>(1); run<4, 5>(1); run<4, 7>(1); run<1, 7>(1);
  run<1, 0>(2); run<4, 5>(2); run<2, 5>(2); run<4, 7>(2); run<4, 5>(3);
  run2<1>(1); run2<2>(1); run2<1>(2); run2<2>(2); run2<1>(3); run2<2>(3); run2<1>(4);
  return 0;
}
