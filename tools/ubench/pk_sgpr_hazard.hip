// Does a packed-f32 VALU instruction (v_pk_mul_f32: 4 passes of 16 lanes on gfx950) see ONE value of an SGPR operand
// for all its lanes when that SGPR is written shortly before (v_readlane, the form an SGPR spill reload takes) or
// rewritten shortly after (s_mov: write after read)?  Found while root-causing a run-to-run difference of
// k_verify_fused: the pose composition of the guided pass, SLP-vectorised into v_pk_*_f32 with
// SGPR-pair operands reloaded by v_readlane right in front of them, came out different in lanes 48-63 of one
// wavefront about once per thousand survivor chains -- only beside wavefronts that keep the matrix pipe busy.
//   mode 0: v_readlane x2, NOPS VALU fillers, v_pk_mul_f32 reading the pair            (read after write)
//   mode 1: v_pk_mul_f32 reading the pair, NOPS fillers, s_mov_b32 rewriting it        (write after read)
// Wavefronts 0..3 of a 1024-thread workgroup (one per SIMD) run the test, the other twelve run back-to-back fp4 MFMAs.
// Prints the number of wrong lanes per 16-lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pk_sgpr_hazard.hip -o tools/ubench/pk_sgpr_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int NOPS>
__global__ void __launch_bounds__(1024) k(unsigned* errs, float* sink, int iters, int with_mfma) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (!with_mfma) return;
    v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0};
    v8i b = {0x2a2a2222, 0x22222a2a, 0x2a2a2222, 0x2a2a2a2a, 0, 0, 0, 0};
    v16f acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
    for (int it = 0; it < iters * 2; ++it) {
      acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 4, 4, 0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 4, 4, 0, 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i];
    sink[threadIdx.x] = t;
    return;
  }
  // the "spill" register: lane L holds the float L + 1
  float spill = (float)(lane + 1);
  v2f x = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const int la = (it & 1) ? 12 : 40, lb = (it & 1) ? 13 : 41;      // the pair's value changes every iteration
    v2f d;
    float tmp = 0.f;
    asm volatile("" : "+v"(tmp));
    if (MODE == 0) {
      if (it & 1)
        asm volatile("v_readlane_b32 s20, %1, 12\n\tv_readlane_b32 s21, %1, 13\n\t"
                     ".rept %3\n\tv_mov_b32 %4, %4\n\t.endr\n\t"
                     "v_pk_mul_f32 %0, s[20:21], %2"
                     : "=&v"(d) : "v"(spill), "v"(x), "i"(NOPS), "v"(tmp) : "s20", "s21");
      else
        asm volatile("v_readlane_b32 s20, %1, 40\n\tv_readlane_b32 s21, %1, 41\n\t"
                     ".rept %3\n\tv_mov_b32 %4, %4\n\t.endr\n\t"
                     "v_pk_mul_f32 %0, s[20:21], %2"
                     : "=&v"(d) : "v"(spill), "v"(x), "i"(NOPS), "v"(tmp) : "s20", "s21");
    } else {
      // the pair holds this iteration's value well ahead; right behind the packed op a scalar move rewrites it
      if (it & 1)
        asm volatile("v_readlane_b32 s20, %1, 12\n\tv_readlane_b32 s21, %1, 13\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 %0, s[20:21], %2\n\t"
                     ".rept %3\n\tv_mov_b32 %4, %4\n\t.endr\n\t"
                     "s_mov_b32 s20, 0x7fc00000\n\ts_mov_b32 s21, 0x7fc00000"
                     : "=&v"(d) : "v"(spill), "v"(x), "i"(NOPS), "v"(tmp) : "s20", "s21");
      else
        asm volatile("v_readlane_b32 s20, %1, 40\n\tv_readlane_b32 s21, %1, 41\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 %0, s[20:21], %2\n\t"
                     ".rept %3\n\tv_mov_b32 %4, %4\n\t.endr\n\t"
                     "s_mov_b32 s20, 0x7fc00000\n\ts_mov_b32 s21, 0x7fc00000"
                     : "=&v"(d) : "v"(spill), "v"(x), "i"(NOPS), "v"(tmp) : "s20", "s21");
    }
    const float wa = (float)(la + 1) * x[0], wb = (float)(lb + 1) * x[1];
    bad += (d[0] != wa) ? 1u : 0u;
    bad += (d[1] != wb) ? 0x10000u : 0u;
  }
  errs[(blockIdx.x * 4 + wave) * 64 + lane] = bad;
}

template <int MODE, int NOPS>
void run(int with_mfma) {
  unsigned* d; float* sink;
  const int blocks = 256;
  (void)hipMalloc(&d, blocks * 256 * 4); (void)hipMalloc(&sink, 1024 * 4);
  (void)hipMemset(d, 0, blocks * 256 * 4);
  hipLaunchKernelGGL((k<MODE, NOPS>), dim3(blocks), dim3(1024), 0, 0, d, sink, 20000, with_mfma);
  (void)hipDeviceSynchronize();
  unsigned* h = (unsigned*)malloc(blocks * 256 * 4);
  (void)hipMemcpy(h, d, blocks * 256 * 4, hipMemcpyDeviceToHost);
  unsigned long long lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
  for (int i = 0; i < blocks * 256; ++i) { lo[(i & 63) >> 4] += h[i] & 0xFFFF; hi[(i & 63) >> 4] += h[i] >> 16; }
  printf("mode %d (%s), %d fillers, matrix pipe %s: wrong low elements per lane quarter %llu %llu %llu %llu, high %llu %llu %llu %llu\n",
         MODE, MODE == 0 ? "pair written just before" : "pair rewritten just after", NOPS, with_mfma ? "busy" : "idle",
         lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
  free(h); (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  for (int m = 0; m < 2; ++m) {
    run<0, 0>(m); run<0, 1>(m); run<0, 2>(m); run<0, 4>(m);
    run<1, 0>(m); run<1, 1>(m); run<1, 2>(m); run<1, 4>(m);
  }
  return 0;
}
