// Which dynamic-LDS sizes still let N workgroups share a CU (hipOccupancyMaxActiveBlocksPerMultiprocessor on a trivial
// 128-thread kernel): the allocation granule of the 160 KB, i.e. the budget a chain's working set has to meet for 5 / 6 / 7
// chains per CU.  build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_granule.hip -o /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(128) k(int* o) { extern __shared__ int s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[(threadIdx.x + 1) & 127]; }
int main() {
  int prev = -1;
  for (int b = 16 * 1024; b <= 64 * 1024; b += 64) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 128, b) != hipSuccess) { printf("query failed at %d\n", b); return 1; }
    if (n != prev) { printf("from %6d B: %d workgroups per CU\n", b, n); prev = n; }
  }
  return 0;
}
