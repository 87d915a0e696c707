// Round 5's last probe for the position-dependent pose of round 3 -- the variant the round-4
// review asked for: the packed-f32 instruction of the symptom,
//     v_pk_mul_f32 v[..], s[20:21], v[..] op_sel_hi:[0,1]     ; s20 broadcast to both halves
// with its SGPR operand WRITTEN BY v_readlane right in front of it (0..2 instructions) and REWRITTEN by the scalar unit
// right behind it (0..3 instructions), executed by ONE wavefront per SIMD while its THREE sibling wavefronts on the same
// SIMD issue the matcher's instruction mix (v_mfma_scale_f32_32x32x64_f8f6f4 chains + the v_max3 / v_med3 top-2 stream
// + LDS reads): the four 16-lane passes of the packed instruction are interleaved with the siblings' issue exactly as
// in k_verify_fused at four wavefronts per SIMD.  100 KB of LDS per 1024-thread workgroup pins one workgroup = 16
// wavefronts = 4 per SIMD on every CU.  Prints wrong lanes per quarter of the wavefront (the symptom: lanes 48-63, low
// half).  Rounds 3-4 (pk_sgpr_hazard.hip, pk_opsel_war.hip) tested the two hazards separately, beside pure-MFMA siblings.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pk_siblings.hip -o tools/ubench/pk_siblings
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PRE, int POST>
__global__ void __launch_bounds__(1024) k(unsigned* errs, float* sink, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = (float)(i & 255);
  __syncthreads();
  if ((wave & 3) != 0) {
    // sibling: the matcher's mix -- 8 MFMAs, 40 max3 / med3, LDS reads, per trip
    v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, b = a;
    float best = -1e30f, second = -1e30f;
    for (int it = 0; it < iters; ++it) {
      v16f acc0, acc1;
      const float c = lds[(it * 64 + lane) & 4095];
      for (int i = 0; i < 16; ++i) { acc0[i] = c; acc1[i] = c + 1.f; }
      a[0] ^= it;
      for (int kk = 0; kk < 4; ++kk) {
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 4, 4, 0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 4, 4, 0, 0, 0, 0);
      }
      for (int i = 0; i < 16; i += 2) {
        const float x0 = __builtin_amdgcn_fmed3f(best, acc0[i], acc0[i + 1]);
        best = fmaxf(best, fmaxf(acc0[i], acc0[i + 1]));
        second = fmaxf(second, x0);
        const float x1 = __builtin_amdgcn_fmed3f(best, acc1[i], acc1[i + 1]);
        best = fmaxf(best, fmaxf(acc1[i], acc1[i + 1]));
        second = fmaxf(second, x1);
      }
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = best + second;
    return;
  }
  v2f x = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
  float three = 3.f, big = 1024.f;
  asm volatile("" : "+v"(three), "+v"(big));
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    v2f d;
    asm volatile("v_readlane_b32 s20, %2, 0\n\tv_readlane_b32 s21, %3, 0\n\t"
                 ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                 "v_pk_mul_f32 %0, s[20:21], %1 op_sel_hi:[0,1]\n\t"
                 ".rept %5\n\ts_nop 0\n\t.endr\n\t"
                 "s_lshl_b32 s20, s21, 4\n\t"
                 : "=&v"(d) : "v"(x), "v"(three), "v"(big), "i"(PRE), "i"(POST) : "s20", "s21");
    bad_lo += d[0] != 3.f * x[0];
    bad_hi += d[1] != 3.f * x[1];
  }
  if (bad_lo) atomicAdd(&errs[(lane / 16) * 2], bad_lo);
  if (bad_hi) atomicAdd(&errs[(lane / 16) * 2 + 1], bad_hi);
}

template <int PRE, int POST>
static void run(unsigned* d_errs, float* d_sink) {
  hipMemset(d_errs, 0, 8 * 4);
  hipFuncSetAttribute((const void*)k<PRE, POST>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((k<PRE, POST>), dim3(1024), dim3(1024), 100 * 1024, 0, d_errs, d_sink, 20000);
  unsigned h[8];
  hipMemcpy(h, d_errs, sizeof(h), hipMemcpyDeviceToHost);
  printf("v_readlane %d in front, rewrite %d behind, 3 matcher-mix siblings per SIMD, 1024 workgroups x 4 test wavefronts x 20000 trials:"
         " wrong (low, high) per quarter:", PRE, POST);
  for (int g = 0; g < 4; ++g) printf("  lanes %2d-%2d (%u, %u)", 16 * g, 16 * g + 15, h[g * 2], h[g * 2 + 1]);
  printf("\n");
}

int main() {
  unsigned* d_errs; float* d_sink;
  hipMalloc(&d_errs, 8 * 4); hipMalloc(&d_sink, 1024 * 1024 * 4);
  run<0, 0>(d_errs, d_sink); run<0, 1>(d_errs, d_sink); run<1, 0>(d_errs, d_sink); run<1, 1>(d_errs, d_sink);
  run<2, 0>(d_errs, d_sink); run<0, 2>(d_errs, d_sink); run<2, 3>(d_errs, d_sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  return 0;
}
