// fp_case.hip -- which spelling of sqrt(x / y) in float is IEEE-exact on the device? (diagnostic)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k1(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = sqrtf(x[i] / y[i]); }
__global__ void k2(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = __fsqrt_rn(__fdiv_rn(x[i], y[i])); }
__global__ void k3(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = sqrtf(__fdiv_rn(x[i], y[i])); }
__global__ void k4(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = (float)sqrt((double)(x[i] / y[i])); }
__global__ void k5(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = __fsqrt_rn(x[i]); }
__global__ void k6(const float* x, const float* y, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = sqrtf(x[i]); }
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), y(n), o(n);
  srand(3);
  for (int i = 0; i < n; ++i) { x[i] = (float)(rand() % 100000) * 0.001f + 0.001f; y[i] = (float)(1 + rand() % 300); }
  x[0] = 0x1.696954p+6f; y[0] = 55.f;
  float *dx, *dy, *dout;
  (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&dout, n * 4);
  (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
  const char* names[6] = {"sqrtf(x / y)", "__fsqrt_rn(__fdiv_rn(x, y))", "sqrtf(__fdiv_rn(x, y))", "(float)sqrt((double)(x / y))", "__fsqrt_rn(x)", "sqrtf(x)"};
  for (int v = 0; v < 6; ++v) {
    if (v == 0) hipLaunchKernelGGL(k1, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    if (v == 1) hipLaunchKernelGGL(k2, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    if (v == 2) hipLaunchKernelGGL(k3, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    if (v == 3) hipLaunchKernelGGL(k4, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    if (v == 4) hipLaunchKernelGGL(k5, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    if (v == 5) hipLaunchKernelGGL(k6, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
    (void)hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
      volatile float d = x[i] / y[i];
      volatile float s = v < 4 ? sqrtf(d) : sqrtf(x[i]);
      bad += o[i] != s;
    }
    printf("%-32s mismatches %ld of %d\n", names[v], bad, n);
  }
  return 0;
}
