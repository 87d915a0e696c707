// f64_latency.hip -- latency of dependent fp64 operations on one wavefront (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k(const double* in, double* out, long long* cyc) {
  double x = in[threadIdx.x], y = in[64 + threadIdx.x];
  const int N = 512;
  long long t0 = wall_clock64();
  double a = x;
  for (int i = 0; i < N; ++i) a = y / (a + 1.0);
  long long t1 = wall_clock64();
  double b = x;
  for (int i = 0; i < N; ++i) b = sqrt(b + y);
  long long t2 = wall_clock64();
  double c = x;
  for (int i = 0; i < N; ++i) c = fma(c, y, 0.5);
  long long t3 = wall_clock64();
  double d = x;
  for (int i = 0; i < N; ++i) d = d * y + 0.5;
  long long t4 = wall_clock64();
  float e = (float)x, ey = (float)y;
  for (int i = 0; i < N; ++i) e = fmaf(e, ey, 0.5f);
  long long t5 = wall_clock64();
  out[threadIdx.x] = a + b + c + d + e;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.5 + 0.001 * i;
  double *din, *dout; long long* dc;
  (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, 512); (void)hipMalloc(&dc, 64);
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc);
  long long c[5]; (void)hipMemcpy(c, dc, 40, hipMemcpyDeviceToHost);
  const char* n[5] = {"f64 div + add", "f64 sqrt + add", "f64 fma", "f64 mul + add (no contraction)", "f32 fma"};
  for (int i = 0; i < 5; ++i) printf("%-32s %.1f ns per dependent op (100 MHz wall clock, 512 ops)\n", n[i], (double)c[i] * 10.0 / 512);
  return 0;
}
