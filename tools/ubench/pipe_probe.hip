// Which streams dispatch beside each other?  A launch whose workgroups do not all fit on the chip keeps its hardware
// queue's dispatcher busy until the last workgroup has been placed; the question is what else that blocks.  For every
// ordered pair (X, Y) of N streams: a "hog" (8192 workgroups of 1024 threads holding 64 KB of LDS each, ~20 us of sleep
// per workgroup: several hundred microseconds of dispatching) goes to X, a one-wavefront kernel to Y right behind it; the
// table prints how long after the hog's start the small kernel finished, as a fraction of the hog's own duration.
// ~0: Y dispatched beside X.  ~1: Y's kernel waited for X's dispatch to end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/pipe_probe.hip -o tools/ubench/pipe_probe
//   usage: pipe_probe [n_streams=12] [priority: 0 normal, 1 highest]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k_hog(int* sink, int spins) {
  __shared__ int pad[16384];
  pad[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
  __syncthreads();
  if (pad[(threadIdx.x + 1) & 1023] == -1) sink[0] = 1;
}
__global__ void k_tiny(int* sink) { if (threadIdx.x == 999) sink[1] = 1; }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int hi = argc > 2 ? atoi(argv[2]) : 0;
  int least = 0, greatest = 0;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  std::vector<hipStream_t> s(n);
  for (int i = 0; i < n; ++i) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, hi ? greatest : 0));
  int* d; CK(hipMalloc(&d, 64));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  // touch every stream once (the runtime creates a stream's hardware queue at its first use), in index order
  for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[i], d); CK(hipStreamSynchronize(s[i])); }
  printf("%d streams, priority %s (range %d..%d); rows: hog on X, columns: small kernel on Y; entry = (end of small - start of hog) / hog duration\n",
         n, hi ? "highest" : "normal", least, greatest);
  printf("      ");
  for (int y = 0; y < n; ++y) printf("  Y%-3d", y);
  printf("   hog us\n");
  for (int x = 0; x < n; ++x) {
    printf("X%-3d  ", x);
    float hog_us = 0;
    for (int y = 0; y < n; ++y) {
      if (x == y) { printf("    - "); continue; }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s[x]));
      hipLaunchKernelGGL(k_hog, dim3(8192), dim3(1024), 0, s[x], d, 12);
      CK(hipEventRecord(e1, s[x]));
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[y], d);
      CK(hipEventRecord(e2, s[y]));
      CK(hipDeviceSynchronize());
      float t_h = 0, t_y = 0;
      CK(hipEventElapsedTime(&t_h, e0, e1));
      CK(hipEventElapsedTime(&t_y, e0, e2));
      hog_us = t_h * 1e3f;
      printf(" %5.2f", t_y / t_h);
    }
    printf("   %6.0f\n", hog_us);
  }
  return 0;
}
