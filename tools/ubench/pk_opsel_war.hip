// Round 4's reproducer attempt for the position-dependent pose.  The ISA of the pre-fix build
// (tools/pk_isa_scan.py --pre-fix) shows the pose composition of the guided pass as
//     v_pk_mul_f32 v[34:35], s[20:21], v[34:35] op_sel_hi:[0,1]      ; s20 broadcast to BOTH halves
//     s_lshl_b32   s20, s94, 4                                        ; ... and rewritten by the very next instruction
// i.e. a packed-f32 instruction (four passes of 16 lanes on gfx950) whose SGPR operand is overwritten by the scalar unit
// one issue slot later.  Round 3's probe (pk_sgpr_hazard.hip) tested the plain pair form (op_sel_hi:[1,1]) only; this
// one tests the broadcast form, with 0..3 fillers between the read and the rewrite, alone and beside wavefronts that
// keep the fp4 matrix pipe busy.  Prints wrong lanes per 16-lane quarter, low / high half separately.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pk_opsel_war.hip -o tools/ubench/pk_opsel_war
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NOPS, int FORM>      // FORM 0: op_sel_hi:[0,1] (broadcast low), 1: plain pair
__global__ void __launch_bounds__(1024) k(unsigned* errs, float* sink, int iters, int with_mfma) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (!with_mfma) return;
    v8i a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, b = a;
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
  v2f x = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; ++it) {
    v2f d;
    // s20 = 3.0, s21 = 5.0; read by the packed multiply; then s20 (and s21) become 1024.0
    asm volatile("s_mov_b32 s20, 0x40400000\n\ts_mov_b32 s21, 0x40a00000\n\ts_nop 4\n\t"
                 ".if %2 == 0\n\tv_pk_mul_f32 %0, s[20:21], %1 op_sel_hi:[0,1]\n\t.else\n\tv_pk_mul_f32 %0, s[20:21], %1\n\t.endif\n\t"
                 ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                 "s_mov_b32 s20, 0x44800000\n\ts_mov_b32 s21, 0x44800000\n\t"
                 : "=&v"(d) : "v"(x), "i"(FORM), "i"(NOPS) : "s20", "s21");
    const float want_lo = 3.f * x[0], want_hi = (FORM == 0 ? 3.f : 5.f) * x[1];
    bad_lo += d[0] != want_lo;
    bad_hi += d[1] != want_hi;
  }
  if (bad_lo) atomicAdd(&errs[(wave * 4 + lane / 16) * 2], bad_lo);
  if (bad_hi) atomicAdd(&errs[(wave * 4 + lane / 16) * 2 + 1], bad_hi);
}

template <int NOPS, int FORM>
static void run(unsigned* d_errs, float* d_sink, int with_mfma) {
  hipMemset(d_errs, 0, 32 * 4);
  hipLaunchKernelGGL((k<NOPS, FORM>), dim3(1024), dim3(1024), 0, 0, d_errs, d_sink, 20000, with_mfma);
  unsigned h[32];
  hipMemcpy(h, d_errs, sizeof(h), hipMemcpyDeviceToHost);
  unsigned q[4][2] = {};
  for (int w = 0; w < 4; ++w) for (int g = 0; g < 4; ++g) { q[g][0] += h[(w * 4 + g) * 2]; q[g][1] += h[(w * 4 + g) * 2 + 1]; }
  printf("form %s  fillers %d  mfma %d : wrong (low, high) per quarter of the wavefront:", FORM ? "pair     " : "broadcast", NOPS, with_mfma);
  for (int g = 0; g < 4; ++g) printf("  lanes %2d-%2d (%u, %u)", 16 * g, 16 * g + 15, q[g][0], q[g][1]);
  printf("\n");
}

int main() {
  unsigned* d_errs; float* d_sink;
  hipMalloc(&d_errs, 32 * 4); hipMalloc(&d_sink, 1024 * 4);
  for (int mf = 0; mf < 2; ++mf) {
    run<0, 0>(d_errs, d_sink, mf); run<1, 0>(d_errs, d_sink, mf); run<2, 0>(d_errs, d_sink, mf); run<3, 0>(d_errs, d_sink, mf);
    run<0, 1>(d_errs, d_sink, mf); run<1, 1>(d_errs, d_sink, mf);
  }
  return 0;
}
