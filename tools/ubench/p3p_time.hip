// p3p_time.hip -- where do the cycles of one P3P hypothesis go? (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../multi_robot_slam_separators_amd/csrc p3p_time.hip -o p3p_time
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "sf_pnp_math.hpp"

__global__ void __launch_bounds__(64) k(const double* in, float* out, long long* cyc, int reps) {
  const int lane = threadIdx.x;
  double P[3][3], f[3][3], P4[3];
  const double* d = in + lane * 24;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { P[i][j] = d[3 * i + j]; f[i][j] = d[9 + 3 * i + j]; }
  for (int j = 0; j < 3; ++j) P4[j] = d[18 + j];
  const double ox = d[21], oy = d[22];
  float acc = 0.f;
  long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    float coef[12];
    P[0][0] += 1e-9;   // keep the compiler from hoisting
    if (sfd::p3p_best(P, f, P4, ox, oy, 600.0, 600.0, coef)) acc += coef[3] + coef[0];
  }
  long long t1 = wall_clock64();
  // the quartic alone
  double c[5] = {d[0] * 0.1, -d[1], d[2] * 0.3, d[3], 1.0 + fabs(d[4])};
  double rr[4]; bool ok[4];
  long long t2 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    c[0] += 1e-9;
    sfd::quartic_roots(c, rr, ok);
    acc += (float)rr[0];
  }
  long long t3 = wall_clock64();
  out[lane] = acc;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

int main() {
  std::vector<double> in(64 * 24);
  srand(5);
  for (int l = 0; l < 64; ++l) {
    double* d = in.data() + l * 24;
    double Pc[4][3];
    for (int i = 0; i < 4; ++i) { Pc[i][0] = (rand() % 4000) / 1000.0 - 2; Pc[i][1] = (rand() % 4000) / 1000.0 - 2; Pc[i][2] = 2 + (rand() % 8000) / 1000.0; }
    for (int i = 0; i < 3; ++i) {
      double n = sqrt(Pc[i][0] * Pc[i][0] + Pc[i][1] * Pc[i][1] + Pc[i][2] * Pc[i][2]);
      for (int j = 0; j < 3; ++j) { d[3 * i + j] = Pc[i][j] + (j == 0 ? 0.3 : -0.2); d[9 + 3 * i + j] = Pc[i][j] / n; }
    }
    for (int j = 0; j < 3; ++j) d[18 + j] = Pc[3][j] + (j == 0 ? 0.3 : -0.2);
    d[21] = 600.0 * Pc[3][0] / Pc[3][2]; d[22] = 600.0 * Pc[3][1] / Pc[3][2];
  }
  double* din; float* dout; long long* dc;
  (void)hipMalloc(&din, in.size() * 8); (void)hipMalloc(&dout, 256); (void)hipMalloc(&dc, 16);
  (void)hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
  long long c[2];
  (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  printf("wall_clock64 ticks (100 MHz): p3p_best %.1f per call, quartic_roots %.1f per call\n", (double)c[0] / reps, (double)c[1] / reps);
  printf("=> p3p_best %.2f us, quartic_roots %.2f us per wave-call\n", (double)c[0] / reps / 100.0, (double)c[1] / reps / 100.0);
  return 0;
}
