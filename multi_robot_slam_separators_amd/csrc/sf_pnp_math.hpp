// sf_pnp_math.hpp -- device-side numerics of the PnP kernel (k_pnp.hip).
//
// GPU statement of the canonical arithmetic DESIGN.md section 4 fixes for the 3D->2D estimator;
// translation units including it are compiled with -ffp-contract=off.  Every routine uses
// + - * / sqrt fma only, in a fixed order, so the CPU restatement can be compared bit for bit.
//
// Upstream algorithms restated (none of this exists in the reference repository, which calls
// un-vendored rtabmap / OpenCV):
//   minimal solver   P3P, Grunert's quartic (Haralick et al. 1994); OpenCV's RANSAC kernel for
//                    cv::SOLVEPNP_P3P has the same shape: 4-point sample, 4th point picks the root
//   iteration bound  cv::RANSACUpdateNumIters (ptsetreg.cpp)
//   angle            pcl::getAngle3D
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_device_math.hpp"

namespace sfd {

// Four distinct indices in [0,m), m >= 4: the keyed sampler of sample_triplet plus one draw.
__device__ __forceinline__ void sample_quad(uint64_t seed, uint32_t it, uint32_t attempt, uint32_t m,
                                            uint32_t& i0, uint32_t& i1, uint32_t& i2, uint32_t& i3) {
  const uint64_t ha = mix64(seed ^ mix64(((uint64_t)it << 32) | (uint64_t)attempt));
  const uint64_t hb = mix64(ha);
  const uint32_t r0 = (uint32_t)(ha >> 32), r1 = (uint32_t)ha, r2 = (uint32_t)(hb >> 32), r3 = (uint32_t)hb;
  i0 = __umulhi(r0, m);
  i1 = __umulhi(r1, m - 1);
  if (i1 >= i0) ++i1;
  i2 = __umulhi(r2, m - 2);
  const uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
  if (i2 >= lo) ++i2;
  if (i2 >= hi) ++i2;
  i3 = __umulhi(r3, m - 3);
  uint32_t a = lo, b = hi, c = i2;
  if (c < a) { const uint32_t t = a; a = c; c = b; b = t; }
  else if (c < b) { const uint32_t t = b; b = c; c = t; }
  if (i3 >= a) ++i3;
  if (i3 >= b) ++i3;
  if (i3 >= c) ++i3;
}

// Real roots of c[4] x^4 + ... + c[0] in four slots (ok[k] marks a real root): Ferrari's
// factorisation, positive root of the resolvent cubic by bracketed Newton, three Newton polishing
// steps per root.  Slots 0/1 and 2/3 are the root pairs of the two quadratic factors.
__device__ inline void quartic_roots(const double (&c)[5], double (&r)[4], bool (&ok)[4]) {
  ok[0] = ok[1] = ok[2] = ok[3] = false;
  r[0] = r[1] = r[2] = r[3] = 0.0;
  double cmax = 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) { const double a = fabs(c[i]); if (a > cmax) cmax = a; }
  if (!(fabs(c[4]) > 1e-12 * cmax)) return;
  const double inv = 1.0 / c[4];
  const double b = c[3] * inv, cc = c[2] * inv, d = c[1] * inv, e = c[0] * inv;
  const double b2 = b * b;
  const double p = cc - 0.375 * b2;
  const double q = (d - 0.5 * (b * cc)) + 0.125 * (b2 * b);
  const double rr = ((e - 0.25 * (b * d)) + 0.0625 * (b2 * cc)) - 0.01171875 * (b2 * b2);
  if (q == 0.0) {
    const double disc = p * p - 4.0 * rr;
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      const double z1 = 0.5 * (-p + sd), z2 = 0.5 * (-p - sd);
      if (z1 >= 0.0) { const double s = sqrt(z1); r[0] = s; r[1] = -s; ok[0] = ok[1] = true; }
      if (z2 >= 0.0) { const double s = sqrt(z2); r[2] = s; r[3] = -s; ok[2] = ok[3] = true; }
    }
  } else {
    const double g1 = 0.25 * (p * p) - rr, g0 = -0.125 * (q * q);
    // Fujiwara's bound on the roots, 2 max(|p|, |g1|^(1/2), |g0 / 2|^(1/3)), with the cube root
    // replaced by the next power of two above it (frexp / ldexp: exact)
    double hi = fabs(p);
    { const double sg = sqrt(fabs(g1)); if (sg > hi) hi = sg; }
    {
      int ex;
      (void)frexp(0.5 * fabs(g0), &ex);                 // |g0|/2 = f 2^ex, f in [0.5, 1)
      const double cb = ldexp(1.0, ex >= 0 ? (ex + 2) / 3 : -((-ex) / 3));   // 2^ceil(ex/3) >= cbrt
      if (cb > hi) hi = cb;
    }
    hi = 2.0 * hi;
    double lo = 0.0;
    double m = hi;
    for (int it = 0; it < 128; ++it) {
      const double gm = ((m + p) * m + g1) * m + g0;
      const double dg = (3.0 * m + 2.0 * p) * m + g1;
      if (gm > 0.0) hi = m; else lo = m;
      double mn = m - gm / dg;
      if (fabs(mn - m) <= 4e-16 * fabs(m)) break;   // Newton step within two ulps: converged
      if (hi - lo <= 4e-16 * fabs(m)) break;        // bracket collapsed onto the root
      if (!(mn > lo && mn < hi)) mn = 0.5 * (lo + hi);
      m = mn;
    }
    if (!(m > 0.0)) return;
    const double s = sqrt(2.0 * m);
    const double h = q / (2.0 * s);
    const double k = 0.5 * p + m;
    double disc = 2.0 * m - 4.0 * (k + h);
    if (disc >= 0.0) { const double sd = sqrt(disc); r[0] = 0.5 * (s + sd); r[1] = 0.5 * (s - sd); ok[0] = ok[1] = true; }
    disc = 2.0 * m - 4.0 * (k - h);
    if (disc >= 0.0) { const double sd = sqrt(disc); r[2] = 0.5 * (-s + sd); r[3] = 0.5 * (-s - sd); ok[2] = ok[3] = true; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double x = r[i] - 0.25 * b;
    for (int k = 0; k < 3; ++k) {
      const double f = (((x + b) * x + cc) * x + d) * x + e;
      const double df = ((4.0 * x + 3.0 * b) * x + 2.0 * cc) * x + d;
      if (df == 0.0) break;
      const double xn = x - f / df;
      if (!isfinite(xn)) break;
      x = xn;
    }
    r[i] = x;
  }
}

__device__ __forceinline__ double dot3(const double (&a)[3], const double (&b)[3]) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

// Orthonormal frame of the triangle (A, B, C): rows e1 (along AB), e2 = e3 x e1, e3 (normal).
// COMPUTE = true returns the two normalisation factors in inv[]; COMPUTE = false reuses them: the
// camera-frame triangle of a P3P root is congruent to the world triangle, so its frame needs no
// square root or division of its own.
template <bool COMPUTE>
__device__ inline bool tri_frame(const double (&A)[3], const double (&B)[3], const double (&C)[3],
                                 double (&E)[3][3], double (&inv)[2]) {
  double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  if (COMPUTE) {
    const double n1 = dot3(e1, e1);
    if (!(n1 > 0.0)) return false;
    inv[0] = 1.0 / sqrt(n1);
  }
  e1[0] *= inv[0]; e1[1] *= inv[0]; e1[2] *= inv[0];
  const double w[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
  double e3[3] = {e1[1] * w[2] - e1[2] * w[1], e1[2] * w[0] - e1[0] * w[2], e1[0] * w[1] - e1[1] * w[0]};
  if (COMPUTE) {
    const double n3 = dot3(e3, e3);
    if (!(n3 > 0.0)) return false;
    inv[1] = 1.0 / sqrt(n3);
  }
  e3[0] *= inv[1]; e3[1] *= inv[1]; e3[2] *= inv[1];
#pragma unroll
  for (int j = 0; j < 3; ++j) { E[0][j] = e1[j]; E[2][j] = e3[j]; }
  E[1][0] = e3[1] * e1[2] - e3[2] * e1[1];
  E[1][1] = e3[2] * e1[0] - e3[0] * e1[2];
  E[1][2] = e3[0] * e1[1] - e3[1] * e1[0];
  return true;
}

// P3P on (P[0..2], unit bearings f[0..2]); among the admissible roots the pose with the smallest
// squared reprojection error of the fourth correspondence (P4 -> offsets (ox, oy) from the principal
// point, focal lengths fx, fy) is returned as float coefficients (x_cam = coef * x_world).
__device__ inline bool p3p_best(const double (&P)[3][3], const double (&f)[3][3], const double (&P4)[3],
                                double ox, double oy, double fx, double fy, float (&coef)[12]) {
  const double d23[3] = {P[1][0] - P[2][0], P[1][1] - P[2][1], P[1][2] - P[2][2]};
  const double d13[3] = {P[0][0] - P[2][0], P[0][1] - P[2][1], P[0][2] - P[2][2]};
  const double d12[3] = {P[0][0] - P[1][0], P[0][1] - P[1][1], P[0][2] - P[1][2]};
  const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
  if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return false;
  const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);
  const double K = (a2 - c2) / b2, rcb = c2 / b2;
  const double N0 = K + 1.0, N1 = -2.0 * (K * cb), N2 = K - 1.0;
  const double D0 = 2.0 * cg, D1 = -2.0 * ca;
  const double Q1 = -2.0 * cb;
  const double DD0 = D0 * D0, DD1 = 2.0 * (D0 * D1), DD2 = D1 * D1;
  const double NN0 = N0 * N0, NN1 = 2.0 * (N0 * N1), NN2 = 2.0 * (N0 * N2) + N1 * N1, NN3 = 2.0 * (N1 * N2),
               NN4 = N2 * N2;
  const double ND0 = N0 * D0, ND1 = N0 * D1 + N1 * D0, ND2 = N1 * D1 + N2 * D0, ND3 = N2 * D1;
  const double QD0 = DD0, QD1 = DD1 + Q1 * DD0, QD2 = (DD2 + Q1 * DD1) + DD0, QD3 = Q1 * DD2 + DD1, QD4 = DD2;
  const double tw = 2.0 * cg;
  double c[5];
  c[0] = ((DD0 + NN0) - tw * ND0) - rcb * QD0;
  c[1] = ((DD1 + NN1) - tw * ND1) - rcb * QD1;
  c[2] = ((DD2 + NN2) - tw * ND2) - rcb * QD2;
  c[3] = (NN3 - tw * ND3) - rcb * QD3;
  c[4] = NN4 - rcb * QD4;
  double v[4];
  bool ok[4];
  quartic_roots(c, v, ok);
  double E[3][3], finv[2];
  if (!tri_frame<true>(P[0], P[1], P[2], E, finv)) return false;
  bool have = false;
  double best_e = __longlong_as_double(0x7FF0000000000000LL);
  for (int k = 0; k < 4; ++k) {
    // select slot k without dynamic register indexing
    const double vv = k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
    const bool vk = k == 0 ? ok[0] : (k == 1 ? ok[1] : (k == 2 ? ok[2] : ok[3]));
    if (!vk) continue;
    if (!(vv > 0.0)) continue;
    const double den = D0 + D1 * vv;
    if (den == 0.0) continue;
    const double u = ((N2 * vv + N1) * vv + N0) / den;
    if (!(u > 0.0)) continue;
    const double qv = (vv + Q1) * vv + 1.0;
    if (!(qv > 0.0)) continue;
    const double s1 = sqrt(b2 / qv), s2 = u * s1, s3 = vv * s1;
    if (!isfinite(s1) || !isfinite(s2) || !isfinite(s3)) continue;
    const double C1[3] = {s1 * f[0][0], s1 * f[0][1], s1 * f[0][2]};
    const double C2[3] = {s2 * f[1][0], s2 * f[1][1], s2 * f[1][2]};
    const double C3[3] = {s3 * f[2][0], s3 * f[2][1], s3 * f[2][2]};
    double G[3][3];
    tri_frame<false>(C1, C2, C3, G, finv);
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = (G[0][i] * E[0][j] + G[1][i] * E[1][j]) + G[2][i] * E[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = C1[i] - ((R[3 * i] * P[0][0] + R[3 * i + 1] * P[0][1]) + R[3 * i + 2] * P[0][2]);
    const double X = ((R[0] * P4[0] + R[1] * P4[1]) + R[2] * P4[2]) + t[0];
    const double Y = ((R[3] * P4[0] + R[4] * P4[1]) + R[5] * P4[2]) + t[1];
    const double Z = ((R[6] * P4[0] + R[7] * P4[1]) + R[8] * P4[2]) + t[2];
    if (!(Z > 0.0)) continue;
    const double iz = 1.0 / Z;
    const double du = fx * (X * iz) - ox, dv = fy * (Y * iz) - oy;
    const double e = du * du + dv * dv;
    if (e < best_e) {
      best_e = e;
      have = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) coef[4 * i + j] = (float)R[3 * i + j];
        coef[4 * i + 3] = (float)t[i];
      }
    }
  }
  return have;
}

// atan2(y, x) for y >= 0, result in [0, pi]: two half-angle reductions, Maclaurin series to z^23.
__device__ inline double canon_atan2(double y, double x) {
  const double ax = fabs(x);
  if (!(y > 0.0) && !(ax > 0.0)) return 0.0;
  const bool swap = y > ax;
  double z = swap ? ax / y : y / ax;
  z = z / (1.0 + sqrt(1.0 + z * z));
  z = z / (1.0 + sqrt(1.0 + z * z));
  const double z2 = z * z;
  double s = 1.0 / 23.0;
#pragma unroll
  for (int k = 10; k >= 0; --k) s = 1.0 / (double)(2 * k + 1) - z2 * s;
  double a = 4.0 * (z * s);
  if (swap) a = 1.57079632679489661923 - a;
  return x < 0.0 ? 3.14159265358979323846 - a : a;
}

// cv::RANSACUpdateNumIters with the canonical logarithm
__device__ inline int update_num_iters(double p, double ep, int model_points, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
  const double w = 1.0 - ep;
  double wp = 1.0;
  for (int i = 0; i < model_points; ++i) wp = wp * w;
  double denom = 1.0 - wp;
  if (denom < 2.2250738585072014e-308) return 0;
  num = canon_log(num);
  denom = canon_log(denom);
  if (denom >= 0.0 || -num >= (double)max_iters * (-denom)) return max_iters;
  return (int)rint(num / denom);
}

__device__ inline void quat_to_R(const double (&q)[4], double (&R)[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
  const double wx = w * x, wy = w * y, wz = w * z;
  R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
}

__device__ inline void R_to_quat(const double (&R)[9], double (&q)[4]) {
  const double tr = (R[0] + R[4]) + R[8];
  double w, x, y, z;
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    w = 0.5 * s; s = 0.5 / s;
    x = (R[7] - R[5]) * s; y = (R[2] - R[6]) * s; z = (R[3] - R[1]) * s;
  } else if (R[0] >= R[4] && R[0] >= R[8]) {
    double s = sqrt(((R[0] - R[4]) - R[8]) + 1.0);
    x = 0.5 * s; s = 0.5 / s;
    w = (R[7] - R[5]) * s; y = (R[1] + R[3]) * s; z = (R[2] + R[6]) * s;
  } else if (R[4] >= R[8]) {
    double s = sqrt(((R[4] - R[0]) - R[8]) + 1.0);
    y = 0.5 * s; s = 0.5 / s;
    w = (R[2] - R[6]) * s; x = (R[1] + R[3]) * s; z = (R[5] + R[7]) * s;
  } else {
    double s = sqrt(((R[8] - R[0]) - R[4]) + 1.0);
    z = 0.5 * s; s = 0.5 / s;
    w = (R[3] - R[1]) * s; x = (R[2] + R[6]) * s; y = (R[5] + R[7]) * s;
  }
  const double inv = 1.0 / sqrt(((w * w + x * x) + y * y) + z * z);
  q[0] = w * inv; q[1] = x * inv; q[2] = y * inv; q[3] = z * inv;
}

// (H with its diagonal scaled by 1 + lambda) d = -g by Cholesky.  ne[0..20] = upper triangle of H
// (row-major j <= k), ne[21..26] = g.  false when H is not positive definite.
__device__ inline bool solve6(const double* __restrict__ ne, double lambda, double (&d)[6]) {
  double A[6][6], Lm[6][6];
  {
    int o = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int k = j; k < 6; ++k) { A[j][k] = ne[o]; A[k][j] = ne[o]; ++o; }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) A[j][j] = A[j][j] * (1.0 + lambda);
  bool good = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) s = s - Lm[j][k] * Lm[j][k];
    if (!(s > 0.0) || !isfinite(s)) { good = false; s = 1.0; }
    const double ljj = sqrt(s);
    Lm[j][j] = ljj;
    const double inv = 1.0 / ljj;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - Lm[i][k] * Lm[j][k];
      Lm[i][j] = v * inv;
    }
  }
  if (!good) return false;
  double yv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = -ne[21 + i];
#pragma unroll
    for (int k = 0; k < i; ++k) v = v - Lm[i][k] * yv[k];
    yv[i] = v / Lm[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = yv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v = v - Lm[k][i] * d[k];
    d[i] = v / Lm[i][i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) good = good && isfinite(d[i]);
  return good;
}

}  // namespace sfd
