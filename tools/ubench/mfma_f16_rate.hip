// Issue rate of the two fp16 MFMA forms on gfx950 (one wavefront per SIMD, independent accumulators, back to back):
// the CDNA3-era v_mfma_f32_32x32x8_f16 against CDNA4's v_mfma_f32_32x32x16_f16 -- does the old opcode run at the new rate?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_f16_rate.hip -o tools/ubench/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int FORM>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  h4 a4 = {(_Float16)lane, 1, 2, 3}, b4 = {1, (_Float16)lane, 1, 1};
  h8 a8 = {(_Float16)lane, 1, 2, 3, 4, 5, 6, 7}, b8 = {1, (_Float16)lane, 1, 1, 1, 1, 1, 1};
  f16v c0, c1, c2, c3;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 1.f; c2[i] = 2.f; c3[i] = 3.f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (FORM == 8) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float t = 0.f;
  for (int i = 0; i < 16; ++i) t += c0[i] + c1[i] + c2[i] + c3[i];
  sink[threadIdx.x] = t;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int FORM> void run(const char* name) {
  unsigned long long* d; float* sink;
  (void)hipMalloc(&d, 8); (void)hipMalloc(&sink, 1024);
  const int iters = 100000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<FORM>, dim3(1), dim3(256), 0, 0, d, sink, 1000);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<FORM>, dim3(1), dim3(256), 0, 0, d, sink, iters);
  (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / (4.0 * iters);
  printf("%-28s %6.2f ns per MFMA and SIMD = %5.1f cycles at 2.4 GHz; %5.0f flop per ns and SIMD\n", name, ns, ns * 2.4,
         (FORM == 8 ? 16384.0 : 32768.0) / ns);
}
int main() { run<8>("v_mfma_f32_32x32x8_f16"); run<16>("v_mfma_f32_32x32x16_f16"); return 0; }
