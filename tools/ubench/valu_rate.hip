// Micro-benchmark: per-instruction VALU issue rate on gfx950 (lane-ops / s, whole chip).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
  uint32_t s = seed | 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const int j = r & 7;
      if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 2) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 4) asm volatile("v_lshl_or_b32 %0, %1, 16, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 5) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[j]) : "s"(s));
      if (OP == 6) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 8) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(uint64_t*)&a[j & 6]) : "v"(*(uint64_t*)&a[(j + 2) & 6]));
      if (OP == 9) asm volatile("v_and_or_b32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 10) asm volatile("v_min3_u32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 11) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 12) asm volatile("v_med3_u32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 13) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 14) asm volatile("v_med3_f32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 15) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[j]));
      if (OP == 16) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(a[(j + 1) & 7]) : "vcc");
      if (OP == 17) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 18) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[j]));
      if (OP == 19) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 20) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 21) asm volatile("v_sad_u32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 22) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 23) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[j]) : "s"(s));
      if (OP == 24) asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 25) asm volatile("v_min_i32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 26) asm volatile("v_min_u16 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 27) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (OP == 28) asm volatile("v_xor_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %2, %0, %2" : "+v"(a[j]), "+v"(a[(j+1)&7]), "+v"(a[(j+2)&7]) : );
    }
  }
  uint32_t x = 0;
  for (int i = 0; i < 8; ++i) x ^= a[i];
  if (x == 0x12345678u) out[0] = x;
}

template <int OP>
void run(const char* name, uint32_t* d) {
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * REP;
  printf("%-16s %8.2f T lane-ops/s   (%.2f cycles per wave64 instr per SIMD at 2.4 GHz)\n", name, ops / ms / 1e9,
         256.0 * 4 * 2.4e9 * 64 / (ops / (ms * 1e-3)));
}

int main() {
  uint32_t* d; hipMalloc(&d, 64);
  run<0>("v_xor_b32", d); run<1>("v_bcnt_u32_b32", d); run<2>("v_min_u32", d); run<6>("v_max_u32", d);
  run<3>("v_fma_f32", d); run<4>("v_lshl_or_b32", d); run<5>("v_xor_b32 sgpr", d); run<7>("v_add_u32", d);
  run<8>("v_pk_fma_f32", d); run<9>("v_and_or_b32", d); run<10>("v_min3_u32", d); run<11>("v_mad_u32_u24", d);
  run<12>("v_med3_u32", d); run<13>("v_min_f32", d); run<14>("v_med3_f32", d); run<15>("v_cvt_f32_u32", d);
  run<16>("cmp+cndmask(2)", d); run<17>("v_and_b32", d); run<18>("v_lshlrev_b32", d); run<19>("v_add_f32", d);
  run<20>("v_max_f32", d); run<21>("v_sad_u32", d); run<22>("v_dot4_u32_u8", d); run<23>("v_add_u32 sgpr", d);
  run<24>("v_pk_min_u16", d); run<25>("v_min_i32", d); run<26>("v_min_u16", d); run<27>("v_sub_u32", d);
  run<28>("xor+bcnt(2)", d);
  return 0;
}
