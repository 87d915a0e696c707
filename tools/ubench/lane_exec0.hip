// Do v_readlane_b32 / v_writelane_b32 execute when EXEC = 0 on gfx950?  (The ISA manual says both ignore EXEC; LLVM's
// SIInstrInfo::hasUnwantedEffectsWhenEXECEmpty lists them as "operate on undefined data" with EXEC = 0.  The compiler
// places SGPR spill reloads -- v_readlane -- between an s_and_saveexec and its s_cbranch_execz in every large kernel of
// csrc/, i.e. under an EXEC that is 0 in the wavefronts whose lanes all fail the condition.)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lane_exec0.hip -o /tmp/lane_exec0 ; prints four words per test.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned* out) {
  unsigned v = 0x1000u + threadIdx.x;      // lane i holds 0x1000 + i
  unsigned r_exec0 = 0xdeadbeefu, r_exec1 = 0xdeadbeefu, w_exec0 = 0, back = 0xdeadbeefu;
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "s_mov_b64 exec, 0\n\t"
      "s_nop 4\n\t"
      "v_readlane_b32 %[r0], %[v], 5\n\t"          // reload-style read under EXEC = 0
      "s_mov_b32 s20, 0x77\n\t"
      "s_nop 1\n\t"
      "v_writelane_b32 %[v], s20, 9\n\t"           // spill-style write under EXEC = 0
      "s_nop 4\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_nop 4\n\t"
      "v_readlane_b32 %[r1], %[v], 5\n\t"          // the same read with EXEC restored
      "v_readlane_b32 %[bk], %[v], 9\n\t"          // what lane 9 holds now
      : [save] "=&s"(save), [r0] "+s"(r_exec0), [r1] "+s"(r_exec1), [bk] "+s"(back), [v] "+v"(v)
      :
      : "s20", "memory");
  w_exec0 = back;
  if (threadIdx.x == 0) {
    out[0] = r_exec0;   // 0x1005 if the read executed, 0xdeadbeef if it was skipped
    out[1] = r_exec1;   // 0x1005
    out[2] = w_exec0;   // 0x77 if the write executed, 0x1009 if it was skipped
    out[3] = 0;
  }
}

int main() {
  unsigned* d; unsigned h[4];
  hipMalloc(&d, 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("v_readlane under EXEC=0: 0x%x (0x1005 = executed, 0xdeadbeef = skipped); with EXEC restored: 0x%x; "
         "v_writelane under EXEC=0 left lane 9 = 0x%x (0x77 = executed, 0x1009 = skipped)\n", h[0], h[1], h[2]);
  return 0;
}
