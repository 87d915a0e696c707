/*
 * sf_oracle_pnp.c -- CPU ORACLE (test infrastructure, NOT product code; see sf_oracle.h).
 *
 * 3D->2D (PnP) motion estimation, the `_estimationType == 1` branch of the reference:
 * PKG/src/myRegistrationVis.cpp:1055-1112 -> util3d::estimateMotion3DTo2D [upstream rtabmap
 * util3d_motion_estimation.cpp] -> cv::solvePnPRansac [upstream OpenCV calib3d solvepnp.cpp +
 * ptsetreg.cpp RANSACPointSetRegistrator].  PARITY UNPINNED: none of that third-party code is
 * vendored under /root/reference or installed here and the reference holds no golden vectors, so
 * this file restates the published STRUCTURE of those routines
 *     RANSAC over minimal samples -> squared reprojection error <= reprojError^2 ->
 *     adaptive iteration count (confidence 0.99) -> iterative refinement on the inliers ->
 *     rtabmap's frame change and its median-based covariance
 * in the canonical arithmetic of DESIGN.md section 4 (fixed operation order, + - * / sqrt fma
 * only), which the HIP kernel k_pnp restates independently so the two can be compared bit for
 * bit.  Deliberate differences from OpenCV, all documented in DESIGN.md:
 *   - minimal solver: 4-point samples solved by P3P (Grunert's quartic, the 4th point picks the
 *     root) instead of 5-point EPnP -- OpenCV itself uses this kernel for SOLVEPNP_P3P;
 *   - stateless keyed sampler instead of cv::RNG;
 *   - points behind the camera are never inliers (cv::projectPoints has no such test);
 *   - the final solve is Levenberg-Marquardt on the rotation/translation started from the best
 *     RANSAC model (cvFindExtrinsicCameraParams2 starts from a DLT / the extrinsic guess); both
 *     minimise the same reprojection error over the same inlier set.
 */
#include "sf_oracle.h"
#include "sf_oracle_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- keyed sampler: four distinct indices in [0,m), m >= 4 ---------------------------------- */
void sfo_sample_quad(uint64_t seed, uint32_t iteration, uint32_t attempt, uint32_t m, uint32_t out[4]) {
  uint64_t ha = sfo_mix(seed ^ sfo_mix(((uint64_t)iteration << 32) | (uint64_t)attempt));
  uint64_t hb = sfo_mix(ha);
  uint32_t r0 = (uint32_t)(ha >> 32), r1 = (uint32_t)ha, r2 = (uint32_t)(hb >> 32), r3 = (uint32_t)hb;
  uint32_t i0 = (uint32_t)(((uint64_t)r0 * (uint64_t)m) >> 32);
  uint32_t i1 = (uint32_t)(((uint64_t)r1 * (uint64_t)(m - 1)) >> 32);
  if (i1 >= i0) ++i1;
  uint32_t i2 = (uint32_t)(((uint64_t)r2 * (uint64_t)(m - 2)) >> 32);
  uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
  if (i2 >= lo) ++i2;
  if (i2 >= hi) ++i2;
  uint32_t i3 = (uint32_t)(((uint64_t)r3 * (uint64_t)(m - 3)) >> 32);
  /* sorted (a <= b <= c) of the first three */
  uint32_t a = lo, b = hi, c = i2;
  if (c < a) { uint32_t t = a; a = c; c = b; b = t; }
  else if (c < b) { uint32_t t = b; b = c; c = t; }
  if (i3 >= a) ++i3;
  if (i3 >= b) ++i3;
  if (i3 >= c) ++i3;
  out[0] = i0; out[1] = i1; out[2] = i2; out[3] = i3;
}

/* ---- real roots of c[4] x^4 + c[3] x^3 + c[2] x^2 + c[1] x + c[0] ------------------------------
 * Ferrari's factorisation of the depressed quartic into two quadratics.  The positive root of the
 * resolvent cubic is bracketed in [0, Cauchy bound] and found by safeguarded Newton (bisection
 * whenever the Newton step leaves the bracket), so the whole routine needs + - * / sqrt only.
 * Each root is polished by three Newton steps on the monic quartic. */
int sfo_quartic_roots(const double c[5], double r[4]) {
  double cmax = 0.0;
  for (int i = 0; i < 5; ++i) { double a = fabs(c[i]); if (a > cmax) cmax = a; }
  if (!(fabs(c[4]) > 1e-12 * cmax)) return 0;
  const double inv = 1.0 / c[4];
  const double b = c[3] * inv, cc = c[2] * inv, d = c[1] * inv, e = c[0] * inv;
  const double b2 = b * b;
  const double p = cc - 0.375 * b2;
  const double q = (d - 0.5 * (b * cc)) + 0.125 * (b2 * b);
  const double rr = ((e - 0.25 * (b * d)) + 0.0625 * (b2 * cc)) - 0.01171875 * (b2 * b2);
  double y[4];
  int n = 0;
  if (q == 0.0) {
    const double disc = p * p - 4.0 * rr;
    if (disc >= 0.0) {
      const double sd = sqrt(disc);
      const double z1 = 0.5 * (-p + sd), z2 = 0.5 * (-p - sd);
      if (z1 >= 0.0) { const double s = sqrt(z1); y[n++] = s; y[n++] = -s; }
      if (z2 >= 0.0) { const double s = sqrt(z2); y[n++] = s; y[n++] = -s; }
    }
  } else {
    const double g1 = 0.25 * (p * p) - rr, g0 = -0.125 * (q * q);
    /* Fujiwara's bound on the roots, 2 max(|p|, |g1|^(1/2), |g0 / 2|^(1/3)), with the cube root
     * replaced by the next power of two above it (frexp / ldexp: exact) */
    double hi = fabs(p);
    { const double sg = sqrt(fabs(g1)); if (sg > hi) hi = sg; }
    {
      int ex;
      (void)frexp(0.5 * fabs(g0), &ex);                 /* |g0|/2 = f 2^ex, f in [0.5, 1) */
      const double cb = ldexp(1.0, ex >= 0 ? (ex + 2) / 3 : -((-ex) / 3));   /* 2^ceil(ex/3) >= cbrt */
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
      if (fabs(mn - m) <= 4e-16 * fabs(m)) break;   /* Newton step within two ulps: converged */
      if (hi - lo <= 4e-16 * fabs(m)) break;        /* bracket collapsed onto the root */
      if (!(mn > lo && mn < hi)) mn = 0.5 * (lo + hi);
      m = mn;
    }
    if (!(m > 0.0)) return 0;
    const double s = sqrt(2.0 * m);
    const double h = q / (2.0 * s);
    const double k = 0.5 * p + m;
    double disc = 2.0 * m - 4.0 * (k + h);
    if (disc >= 0.0) { const double sd = sqrt(disc); y[n++] = 0.5 * (s + sd); y[n++] = 0.5 * (s - sd); }
    disc = 2.0 * m - 4.0 * (k - h);
    if (disc >= 0.0) { const double sd = sqrt(disc); y[n++] = 0.5 * (-s + sd); y[n++] = 0.5 * (-s - sd); }
  }
  for (int i = 0; i < n; ++i) {
    double x = y[i] - 0.25 * b;
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
  return n;
}

static double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* orthonormal frame of the triangle (A, B, C): e1 along AB, e3 normal, e2 = e3 x e1.
 * inv[0] = 1/|AB|, inv[1] = 1/|AB x AC|/|AB| are RETURNED when compute != 0 and REUSED otherwise:
 * the camera-frame triangle of a P3P root is congruent to the world triangle, so its frame is
 * normalised with the world triangle's factors (no square root or division per root). */
static int sfo_tri_frame(const double* A, const double* B, const double* C, double E[3][3], double inv[2],
                         int compute) {
  double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  if (compute) {
    const double n1 = dot3(e1, e1);
    if (!(n1 > 0.0)) return 0;
    inv[0] = 1.0 / sqrt(n1);
  }
  e1[0] *= inv[0]; e1[1] *= inv[0]; e1[2] *= inv[0];
  const double w[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
  double e3[3] = {e1[1] * w[2] - e1[2] * w[1], e1[2] * w[0] - e1[0] * w[2], e1[0] * w[1] - e1[1] * w[0]};
  if (compute) {
    const double n3 = dot3(e3, e3);
    if (!(n3 > 0.0)) return 0;
    inv[1] = 1.0 / sqrt(n3);
  }
  e3[0] *= inv[1]; e3[1] *= inv[1]; e3[2] *= inv[1];
  for (int j = 0; j < 3; ++j) { E[0][j] = e1[j]; E[2][j] = e3[j]; }
  E[1][0] = e3[1] * e1[2] - e3[2] * e1[1];
  E[1][1] = e3[2] * e1[0] - e3[0] * e1[2];
  E[1][2] = e3[0] * e1[1] - e3[1] * e1[0];
  return 1;
}

/* ---- P3P (Grunert 1841, in the form of Haralick et al. 1994) -------------------------------------
 * World points P[0..2], unit bearing vectors f[0..2] in the camera frame.  With depths s1, s2 = u s1,
 * s3 = v s1 the three cosine-law equations reduce to a quartic in v whose coefficients are built
 * here by explicit polynomial products (no memorised closed forms):
 *     u(v) = N(v) / D(v),   N = (K-1) v^2 - 2 K cos(b) v + (K+1),  D = 2 (cos(g) - v cos(a)),
 *     0 = D^2 + N^2 - 2 cos(g) N D - (c^2/b^2) (1 + v^2 - 2 v cos(b)) D^2,     K = (a^2 - c^2)/b^2.
 * Each admissible root gives camera-frame points s_i f_i; the pose aligns the two triangles through
 * their orthonormal frames.  Returns the number of poses (x_cam = R x_world + t). */
int sfo_p3p(const double P[3][3], const double f[3][3], double R[4][9], double t[4][3]) {
  const double d23[3] = {P[1][0] - P[2][0], P[1][1] - P[2][1], P[1][2] - P[2][2]};
  const double d13[3] = {P[0][0] - P[2][0], P[0][1] - P[2][1], P[0][2] - P[2][2]};
  const double d12[3] = {P[0][0] - P[1][0], P[0][1] - P[1][1], P[0][2] - P[1][2]};
  const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
  if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
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
  const int nr = sfo_quartic_roots(c, v);
  double E[3][3], finv[2];
  if (!sfo_tri_frame(P[0], P[1], P[2], E, finv, 1)) return 0;
  int ns = 0;
  for (int k = 0; k < nr; ++k) {
    const double vv = v[k];
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
    sfo_tri_frame(C1, C2, C3, G, finv, 0);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[ns][3 * i + j] = (G[0][i] * E[0][j] + G[1][i] * E[1][j]) + G[2][i] * E[2][j];
    for (int i = 0; i < 3; ++i)
      t[ns][i] = C1[i] - ((R[ns][3 * i] * P[0][0] + R[ns][3 * i + 1] * P[0][1]) + R[ns][3 * i + 2] * P[0][2]);
    ++ns;
  }
  return ns;
}

/* ---- canonical atan2(y, x) for y >= 0: result in [0, pi] ------------------------------------------
 * two half-angle reductions z <- z / (1 + sqrt(1 + z^2)) bring the argument under tan(pi/16), then
 * the Maclaurin series to z^23. */
double sfo_canon_atan2(double y, double x) {
  const double ax = fabs(x);
  if (!(y > 0.0) && !(ax > 0.0)) return 0.0;
  const int swap = y > ax;
  double z = swap ? ax / y : y / ax;
  z = z / (1.0 + sqrt(1.0 + z * z));
  z = z / (1.0 + sqrt(1.0 + z * z));
  const double z2 = z * z;
  double s = 1.0 / 23.0;
  for (int k = 10; k >= 0; --k) s = 1.0 / (double)(2 * k + 1) - z2 * s;
  double a = 4.0 * (z * s);
  if (swap) a = 1.57079632679489661923 - a;
  return x < 0.0 ? 3.14159265358979323846 - a : a;
}

/* cv::RANSACUpdateNumIters [upstream OpenCV ptsetreg.cpp] with the canonical logarithm */
static int sfo_update_num_iters(double p, double ep, int model_points, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  double w = 1.0 - ep, wp = 1.0;
  for (int i = 0; i < model_points; ++i) wp = wp * w;
  double denom = 1.0 - wp;
  if (denom < DBL_MIN) return 0;
  num = sfo_canon_log(num);
  denom = sfo_canon_log(denom);
  if (denom >= 0.0 || -num >= (double)max_iters * (-denom)) return max_iters;
  return (int)nearbyint(num / denom);   /* cvRound: round half to even */
}

typedef struct {
  int m;
  const float* obj;     /* m x 3 world ("from" base frame) points  */
  const float* dpx;     /* m     pixel x - cx (float)               */
  const float* dpy;     /* m     pixel y - cy                       */
  float fxf, fyf, thr2f;
  double fx, fy;
} sfo_pnp_problem;

/* one correspondence against float coefficients c[12] (x_cam = c * x_world): canonical fma chain,
 * division-free squared reprojection error test  A^2 + B^2 <= thr^2 Z^2,  Z > 0 */
static int sfo_pnp_inlier(const sfo_pnp_problem* pb, const float c[12], int i) {
  const float* P = pb->obj + 3 * i;
  const float X = fmaf(c[2], P[2], fmaf(c[1], P[1], fmaf(c[0], P[0], c[3])));
  const float Y = fmaf(c[6], P[2], fmaf(c[5], P[1], fmaf(c[4], P[0], c[7])));
  const float Z = fmaf(c[10], P[2], fmaf(c[9], P[1], fmaf(c[8], P[0], c[11])));
  const float A = fmaf(-pb->dpx[i], Z, pb->fxf * X);
  const float B = fmaf(-pb->dpy[i], Z, pb->fyf * Y);
  const float lhs = fmaf(B, B, A * A);
  const float rhs = pb->thr2f * (Z * Z);
  return (Z > 0.0f) && (lhs <= rhs);
}

/* hypothesis of one RANSAC iteration: sample four, P3P on the first three, the fourth picks the
 * root (smallest squared reprojection error, first on ties).  Returns 1 and the float model. */
static int sfo_pnp_hypothesis(const sfo_pnp_problem* pb, uint64_t seed, uint32_t it, float coef[12]) {
  uint32_t s[4];
  sfo_sample_quad(seed, it, 0u, (uint32_t)pb->m, s);
  double P[3][3], f[3][3];
  for (int k = 0; k < 3; ++k) {
    for (int j = 0; j < 3; ++j) P[k][j] = (double)pb->obj[3 * s[k] + j];
    const double un = (double)pb->dpx[s[k]] / pb->fx, vn = (double)pb->dpy[s[k]] / pb->fy;
    const double inv = 1.0 / sqrt((un * un + vn * vn) + 1.0);
    f[k][0] = un * inv; f[k][1] = vn * inv; f[k][2] = inv;
  }
  double R[4][9], t[4][3];
  const int ns = sfo_p3p(P, f, R, t);
  const double P4[3] = {(double)pb->obj[3 * s[3]], (double)pb->obj[3 * s[3] + 1], (double)pb->obj[3 * s[3] + 2]};
  const double ox = (double)pb->dpx[s[3]], oy = (double)pb->dpy[s[3]];
  int best = -1;
  double best_e = INFINITY;
  for (int k = 0; k < ns; ++k) {
    const double X = ((R[k][0] * P4[0] + R[k][1] * P4[1]) + R[k][2] * P4[2]) + t[k][0];
    const double Y = ((R[k][3] * P4[0] + R[k][4] * P4[1]) + R[k][5] * P4[2]) + t[k][1];
    const double Z = ((R[k][6] * P4[0] + R[k][7] * P4[1]) + R[k][8] * P4[2]) + t[k][2];
    if (!(Z > 0.0)) continue;
    const double iz = 1.0 / Z;
    const double du = pb->fx * (X * iz) - ox, dv = pb->fy * (Y * iz) - oy;
    const double e = du * du + dv * dv;
    if (e < best_e) { best_e = e; best = k; }
  }
  if (best < 0) return 0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) coef[4 * i + j] = (float)R[best][3 * i + j];
    coef[4 * i + 3] = (float)t[best][i];
  }
  return 1;
}

void sfo_quat_to_R(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
  const double wx = w * x, wy = w * y, wz = w * z;
  R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
}

/* Shepperd's rotation-matrix -> unit quaternion (w, x, y, z) */
void sfo_R_to_quat(const double R[9], double q[4]) {
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

/* Normal equations of the reprojection error over the members of `mask` at pose (q, t):
 * out[0..20] = upper triangle of J^T J (row-major j <= k), out[21..26] = J^T r, out[27] = sum |r|^2.
 * Parameters: left rotation perturbation of R x (3) and translation (3).  Block-order sums. */
#define SFO_PNP_NSUM 28
static void sfo_pnp_normal_eq(const sfo_pnp_problem* pb, const uint8_t* mask, const double q[4],
                              const double t[3], double out[SFO_PNP_NSUM], double* scratch /* 28 * m */) {
  double R[9];
  sfo_quat_to_R(q, R);
  const int m = pb->m;
  for (int i = 0; i < m; ++i) {
    double term[SFO_PNP_NSUM];
    for (int k = 0; k < SFO_PNP_NSUM; ++k) term[k] = 0.0;
    if (mask[i]) {
      const double Px = (double)pb->obj[3 * i], Py = (double)pb->obj[3 * i + 1], Pz = (double)pb->obj[3 * i + 2];
      const double Yx = (R[0] * Px + R[1] * Py) + R[2] * Pz;
      const double Yy = (R[3] * Px + R[4] * Py) + R[5] * Pz;
      const double Yz = (R[6] * Px + R[7] * Py) + R[8] * Pz;
      const double X = Yx + t[0], Y = Yy + t[1], Z = Yz + t[2];
      if (Z > 0.0) {
        const double iz = 1.0 / Z;
        const double xn = X * iz, yn = Y * iz;
        const double ru = pb->fx * xn - (double)pb->dpx[i];
        const double rv = pb->fy * yn - (double)pb->dpy[i];
        const double a0 = pb->fx * iz, a2 = -(a0 * xn);
        const double b1 = pb->fy * iz, b2 = -(b1 * yn);
        double Ju[6], Jv[6];
        Ju[0] = a2 * Yy;            Ju[1] = a0 * Yz - a2 * Yx; Ju[2] = -(a0 * Yy);
        Ju[3] = a0;                 Ju[4] = 0.0;               Ju[5] = a2;
        Jv[0] = b2 * Yy - b1 * Yz;  Jv[1] = -(b2 * Yx);        Jv[2] = b1 * Yx;
        Jv[3] = 0.0;                Jv[4] = b1;                Jv[5] = b2;
        int o = 0;
        for (int j = 0; j < 6; ++j)
          for (int k = j; k < 6; ++k) term[o++] = Ju[j] * Ju[k] + Jv[j] * Jv[k];
        for (int j = 0; j < 6; ++j) term[21 + j] = Ju[j] * ru + Jv[j] * rv;
        term[27] = ru * ru + rv * rv;
      } else {
        term[27] = 1e30;   /* a member behind the camera makes the pose unacceptable */
      }
    }
    for (int k = 0; k < SFO_PNP_NSUM; ++k) scratch[(size_t)k * m + i] = term[k];
  }
  for (int k = 0; k < SFO_PNP_NSUM; ++k) out[k] = sfo_block_sum(scratch + (size_t)k * m, m);
}

/* solve (H with diagonal scaled by 1 + lambda) d = -g by Cholesky; 0 when not positive definite */
int sfo_pnp_solve6(const double ne[SFO_PNP_NSUM], double lambda, double d[6]) {
  double A[6][6], Lm[6][6];
  int o = 0;
  for (int j = 0; j < 6; ++j)
    for (int k = j; k < 6; ++k) { A[j][k] = ne[o]; A[k][j] = ne[o]; ++o; }
  for (int j = 0; j < 6; ++j) A[j][j] = A[j][j] * (1.0 + lambda);
  for (int j = 0; j < 6; ++j) {
    double s = A[j][j];
    for (int k = 0; k < j; ++k) s = s - Lm[j][k] * Lm[j][k];
    if (!(s > 0.0) || !isfinite(s)) return 0;
    const double ljj = sqrt(s);
    Lm[j][j] = ljj;
    const double inv = 1.0 / ljj;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
      for (int k = 0; k < j; ++k) v = v - Lm[i][k] * Lm[j][k];
      Lm[i][j] = v * inv;
    }
  }
  double yv[6];
  for (int i = 0; i < 6; ++i) {
    double v = -ne[21 + i];
    for (int k = 0; k < i; ++k) v = v - Lm[i][k] * yv[k];
    yv[i] = v / Lm[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double v = yv[i];
    for (int k = i + 1; k < 6; ++k) v = v - Lm[k][i] * d[k];
    d[i] = v / Lm[i][i];
  }
  for (int i = 0; i < 6; ++i) if (!isfinite(d[i])) return 0;
  return 1;
}

/* Levenberg-Marquardt on the members of `mask` from pose (q, t); at most 20 evaluations
 * [upstream cvFindExtrinsicCameraParams2: CvLevMarq, 20 iterations, diagonal scaled by 1 + lambda]. */
static int sfo_pnp_refine(const sfo_pnp_problem* pb, const uint8_t* mask, double q[4], double t[3],
                          double* final_err, double* scratch) {
  double ne[SFO_PNP_NSUM], nc[SFO_PNP_NSUM];
  sfo_pnp_normal_eq(pb, mask, q, t, ne, scratch);
  double lambda = 1e-3;
  int evals = 0;
  for (int iter = 0; iter < 20; ++iter) {
    double d[6];
    if (!sfo_pnp_solve6(ne, lambda, d)) {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      continue;
    }
    const double hx = 0.5 * d[0], hy = 0.5 * d[1], hz = 0.5 * d[2];
    const double dn = 1.0 / sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
    const double dw = dn, dx = hx * dn, dy = hy * dn, dz = hz * dn;
    double qc[4], tc[3];
    qc[0] = ((dw * q[0] - dx * q[1]) - dy * q[2]) - dz * q[3];
    qc[1] = ((dw * q[1] + dx * q[0]) + dy * q[3]) - dz * q[2];
    qc[2] = ((dw * q[2] - dx * q[3]) + dy * q[0]) + dz * q[1];
    qc[3] = ((dw * q[3] + dx * q[2]) - dy * q[1]) + dz * q[0];
    const double qn = 1.0 / sqrt(((qc[0] * qc[0] + qc[1] * qc[1]) + qc[2] * qc[2]) + qc[3] * qc[3]);
    for (int i = 0; i < 4; ++i) qc[i] = qc[i] * qn;
    for (int i = 0; i < 3; ++i) tc[i] = t[i] + d[3 + i];
    sfo_pnp_normal_eq(pb, mask, qc, tc, nc, scratch);
    ++evals;
    /* a step below float epsilon relative to the parameters ends the iteration whether or not it
     * lowered the error [upstream CvLevMarq: |param - prevParam| / |prevParam| < FLT_EPSILON] */
    const double dd = ((((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]) + d[4] * d[4]) + d[5] * d[5];
    const double tt = ((tc[0] * tc[0] + tc[1] * tc[1]) + tc[2] * tc[2]) + 1.0;
    if (nc[27] < ne[27]) {
      memcpy(q, qc, sizeof(qc)); memcpy(t, tc, sizeof(tc)); memcpy(ne, nc, sizeof(nc));
      lambda = lambda * 0.1;
      if (lambda < 1e-16) lambda = 1e-16;
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
    }
    if (dd <= 1.4e-14 * tt) break;
  }
  *final_err = ne[27];
  return evals;
}

/* [upstream rtabmap util3d computeReprojErrors] members = points in front of the camera whose
 * reprojection error (pixels, NOT squared) is <= thr under the float-rounded pose; errs[i] receives
 * member i's error.  Canonical fma chain, IEEE float division and sqrt. */
static int sfo_pnp_select(const sfo_pnp_problem* pb, const double q[4], const double t[3], float thr,
                          uint8_t* mask, float* errs) {
  double Rd[9];
  sfo_quat_to_R(q, Rd);
  float c[12];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c[4 * i + j] = (float)Rd[3 * i + j]; c[4 * i + 3] = (float)t[i]; }
  int n = 0;
  for (int i = 0; i < pb->m; ++i) {
    const float* P = pb->obj + 3 * i;
    const float X = fmaf(c[2], P[2], fmaf(c[1], P[1], fmaf(c[0], P[0], c[3])));
    const float Y = fmaf(c[6], P[2], fmaf(c[5], P[1], fmaf(c[4], P[0], c[7])));
    const float Z = fmaf(c[10], P[2], fmaf(c[9], P[1], fmaf(c[8], P[0], c[11])));
    int in = 0;
    if (Z > 0.0f) {
      const float du = fmaf(pb->fxf, X / Z, -pb->dpx[i]);
      const float dv = fmaf(pb->fyf, Y / Z, -pb->dpy[i]);
      const float e = sqrtf(fmaf(dv, dv, du * du));
      in = e <= thr;
      if (in) errs[i] = e;
    }
    mask[i] = (uint8_t)in;
    n += in;
  }
  return n;
}

/* value of rank `rank` (0-based) among v[0..n): only the VALUE matters */
static float sfo_rank_value(float* v, int n, int rank) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    float pivot = v[lo + ((hi - lo) >> 1)];
    int i = lo, j = hi;
    while (i <= j) {
      while (v[i] < pivot) ++i;
      while (v[j] > pivot) --j;
      if (i <= j) { float tmp = v[i]; v[i] = v[j]; v[j] = tmp; ++i; --j; }
    }
    if (rank <= j) hi = j;
    else if (rank >= i) lo = i;
    else break;
  }
  return v[rank];
}

int sfo_estimate_motion_3d2d(const sf_params* p,
                             const float* xyz_from, const sf_keypoint* kp_to, const float* xyz_to /* may be NULL */,
                             const uint16_t* corr_from, const uint16_t* corr_to, int n_corr,
                             sfo_motion* out, uint8_t* inlier_mask_out) {
  memset(out, 0, sizeof(*out));
  out->is_null = 1;
  out->variance = 1.0;       /* *covariance = eye(6) */
  out->variance_ang = 1.0;
  out->ransac_best_iteration = -1;
  if (inlier_mask_out) memset(inlier_mask_out, 0, (size_t)(n_corr > 0 ? n_corr : 0));

  const int cap = n_corr > 0 ? n_corr : 1;
  float* obj = (float*)malloc((size_t)cap * 3 * sizeof(float));
  float* dst3 = (float*)malloc((size_t)cap * 3 * sizeof(float));
  float* dpx = (float*)malloc((size_t)cap * sizeof(float));
  float* dpy = (float*)malloc((size_t)cap * sizeof(float));
  int32_t* orig = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
  uint8_t* mask = (uint8_t*)calloc((size_t)cap, 1);
  uint8_t* mask_b = (uint8_t*)calloc((size_t)cap, 1);
  uint8_t* has3 = (uint8_t*)calloc((size_t)cap, 1);
  float* e1 = (float*)malloc((size_t)cap * sizeof(float));
  float* e2 = (float*)malloc((size_t)cap * sizeof(float));
  double* scratch = (double*)malloc((size_t)cap * SFO_PNP_NSUM * sizeof(double));
  const int max_it = p->iterations > 0 ? p->iterations : 0;
  int32_t* counts = (int32_t*)malloc((size_t)(max_it + 1) * sizeof(int32_t));
  int rc = SF_OK;
  if (!obj || !dst3 || !dpx || !dpy || !orig || !mask || !mask_b || !has3 || !e1 || !e2 || !scratch || !counts) {
    rc = SF_ENOMEM;
    goto done;
  }
  {
    /* [upstream util3d::estimateMotion3DTo2D] ids of words2B found in words3A with a finite 3D
     * point, ascending id */
    const float cxf = (float)p->cx, cyf = (float)p->cy;
    int m = 0;
    for (int i = 0; i < n_corr; ++i) {
      const float* a = xyz_from + 3 * (size_t)corr_from[i];
      if (!sfo_finite3(a)) continue;
      memcpy(obj + 3 * m, a, 12);
      dpx[m] = kp_to[corr_to[i]].x - cxf;
      dpy[m] = kp_to[corr_to[i]].y - cyf;
      if (xyz_to) {
        const float* b = xyz_to + 3 * (size_t)corr_to[i];
        memcpy(dst3 + 3 * m, b, 12);
        has3[m] = (uint8_t)sfo_finite3(b);
      }
      orig[m] = i;
      ++m;
    }
    out->matches = m;
    if (m < p->min_inliers || m < 4) goto done;

    sfo_pnp_problem pb;
    pb.m = m; pb.obj = obj; pb.dpx = dpx; pb.dpy = dpy;
    pb.fx = p->fx; pb.fy = p->fy;
    pb.fxf = (float)p->fx; pb.fyf = (float)p->fy;
    {
      const double thr = (double)p->pnp_reproj_error;
      pb.thr2f = (float)(thr * thr);       /* float t = (float)(thresh*thresh); err <= t */
    }

    /* ---- [upstream cv::RANSACPointSetRegistrator::run] ---------------------------------------- */
    float coef[12];
    for (int it = 0; it < max_it; ++it) {
      int cnt = 0;
      if (sfo_pnp_hypothesis(&pb, p->seed, (uint32_t)it, coef))
        for (int i = 0; i < m; ++i) cnt += sfo_pnp_inlier(&pb, coef, i);
      counts[it] = cnt;
    }
    int niters = max_it, best = 0, best_it = -1, it = 0;
    const int model_points = 4;
    for (it = 0; it < niters; ++it) {
      const int good = counts[it];
      const int bar = best > model_points - 1 ? best : model_points - 1;
      if (good > bar) {
        best = good; best_it = it;
        niters = p->ransac_adaptive_stop
                     ? sfo_update_num_iters(0.99, (double)(m - good) / (double)m, model_points, niters)
                     : niters;
      }
    }
    out->ransac_iterations_run = it;
    out->ransac_best_iteration = best_it;
    out->ransac_best_count = best;
    if (best_it < 0) goto done;      /* solvePnPRansac returns false: no inliers */

    sfo_pnp_hypothesis(&pb, p->seed, (uint32_t)best_it, coef);
    int n_inl = 0;
    for (int i = 0; i < m; ++i) { mask[i] = (uint8_t)sfo_pnp_inlier(&pb, coef, i); n_inl += mask[i]; }

    /* ---- final solve on the inliers ------------------------------------------------------------- */
    double Rb[9], q[4], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Rb[3 * i + j] = (double)coef[4 * i + j]; t[i] = (double)coef[4 * i + 3]; }
    sfo_R_to_quat(Rb, q);
    double err = 0.0;
    out->refine_rounds = sfo_pnp_refine(&pb, mask, q, t, &err, scratch);

    /* ---- [upstream rtabmap util3d::solvePnPRansac] refinement rounds (Vis/PnPRefineIterations > 0):
     * re-solve on the current inliers, re-select with threshold min(reprojError, sigma * stddev of
     * the inlier errors), until the inlier set is stable -- the same loop shape as PCL's refineModel */
    uint8_t* inl = mask;
    const int min_count = p->min_inliers > 4 ? p->min_inliers : 4;
    if (n_inl >= min_count && p->pnp_refine_iterations > 0) {
      const float inlier_thr = p->pnp_reproj_error;
      float error_threshold = inlier_thr;
      int refine_iterations = 0, inlier_changed = 0;
      uint8_t* prev = mask;     /* prev_inliers = inliers */
      uint8_t* neu = mask_b;    /* new_inliers (empty)    */
      int n_prev = n_inl, n_new = 0;
      int sizes[64], n_sizes = 0;
      double qn[4], tn[3];
      memcpy(qn, q, sizeof(qn)); memcpy(tn, t, sizeof(tn));
      do {
        double e_unused;
        out->refine_rounds += sfo_pnp_refine(&pb, prev, qn, tn, &e_unused, scratch);   /* solvePnP, extrinsic guess */
        if (n_sizes < 64) sizes[n_sizes] = n_prev;
        ++n_sizes;
        n_new = sfo_pnp_select(&pb, qn, tn, error_threshold, neu, e1);
        if (n_new < min_count) {
          ++refine_iterations;
          if (refine_iterations >= p->pnp_refine_iterations) break;
          continue;
        }
        /* uMean / uVariance of the inlier errors (block-order sums, float results) */
        for (int i = 0; i < m; ++i) scratch[i] = neu[i] ? (double)e1[i] : 0.0;
        const float mean = (float)(sfo_block_sum(scratch, m) / (double)n_new);
        float variance = 0.0f;
        if (n_new > 1) {
          for (int i = 0; i < m; ++i) {
            const float d = e1[i] - mean;
            scratch[i] = neu[i] ? (double)(d * d) : 0.0;
          }
          variance = (float)(sfo_block_sum(scratch, m) / (double)(n_new - 1));
        }
        const float sthr = (float)p->refine_sigma * sqrtf(variance);
        error_threshold = sthr < inlier_thr ? sthr : inlier_thr;
        inlier_changed = 0;
        { uint8_t* tmp = prev; prev = neu; neu = tmp; int tn_ = n_prev; n_prev = n_new; n_new = tn_; }
        if (n_new != n_prev) {
          if (n_sizes >= 4 && n_sizes <= 64 &&
              sizes[n_sizes - 1] == sizes[n_sizes - 3] && sizes[n_sizes - 2] == sizes[n_sizes - 4])
            break;   /* oscillating */
          inlier_changed = 1;
          continue;
        }
        for (int i = 0; i < m; ++i) if (prev[i] != neu[i]) { inlier_changed = 1; break; }
      } while (inlier_changed && ++refine_iterations < p->pnp_refine_iterations);
      /* std::swap(inliers, new_inliers); rvec = new_model_rvec; tvec = new_model_tvec */
      inl = neu; n_inl = n_new;
      memcpy(q, qn, sizeof(qn)); memcpy(t, tn, sizeof(tn));
    }

    out->inliers = n_inl;
    if (inlier_mask_out) for (int i = 0; i < m; ++i) if (inl[i]) inlier_mask_out[orig[i]] = 1;
    if (n_inl < p->min_inliers) goto done;

    /* ---- transform = (localTransform * pnp).inverse()   (rtabmap::Transform is float) ------------- */
    double Rd[9];
    sfo_quat_to_R(q, Rd);
    float Rf[9], tf[3], MR[9], Mt[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Rd[i];
    for (int i = 0; i < 3; ++i) tf[i] = (float)t[i];
    const float* L = p->local_transform;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        MR[3 * i + j] = (L[4 * i] * Rf[j] + L[4 * i + 1] * Rf[3 + j]) + L[4 * i + 2] * Rf[6 + j];
      Mt[i] = ((L[4 * i] * tf[0] + L[4 * i + 1] * tf[1]) + L[4 * i + 2] * tf[2]) + L[4 * i + 3];
    }
    float* T = out->transform;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T[4 * i + j] = MR[3 * j + i];
      T[4 * i + 3] = -((MR[i] * Mt[0] + MR[3 + i] * Mt[1]) + MR[6 + i] * Mt[2]);
    }
    out->is_null = 0;
    {
      int allz = 1;
      for (int i = 0; i < 12; ++i) if (T[i] != 0.0f) allz = 0;
      if (allz) out->is_null = 1;
    }

    /* ---- covariance [upstream estimateMotion3DTo2D] ------------------------------------------------- */
    if (xyz_to) {
      /* 3D error of the inliers that also have a finite point in the "to" frame:
       * variance = 2.1981 * value at the first quartile (size >> 2) of the sorted errors;
       * linear block from squared distances, angular block from pcl::getAngle3D */
      int oi = 0;
      for (int i = 0; i < m; ++i) {
        if (!inl[i] || !has3[i]) continue;
        const float* b = dst3 + 3 * i;
        const float* a = obj + 3 * i;
        const float nx = fmaf(T[2], b[2], fmaf(T[1], b[1], fmaf(T[0], b[0], T[3])));
        const float ny = fmaf(T[6], b[2], fmaf(T[5], b[1], fmaf(T[4], b[0], T[7])));
        const float nz = fmaf(T[10], b[2], fmaf(T[9], b[1], fmaf(T[8], b[0], T[11])));
        const float dx = nx - a[0], dy = ny - a[1], dz = nz - a[2];
        e1[oi] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const double v1[3] = {(double)(a[0] - T[3]), (double)(a[1] - T[7]), (double)(a[2] - T[11])};
        const double v2[3] = {(double)(nx - T[3]), (double)(ny - T[7]), (double)(nz - T[11])};
        const double cr[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
        e2[oi] = (float)sfo_canon_atan2(sqrt(dot3(cr, cr)), dot3(v1, v2));
        ++oi;
      }
      if (oi > 0) {
        out->variance = 2.1981 * (double)sfo_rank_value(e1, oi, oi >> 2);
        out->variance_ang = 2.1981 * (double)sfo_rank_value(e2, oi, oi >> 2);
      }
    } else {
      /* no 3D in the "to" frame: rms reprojection error of the inliers scales the whole matrix */
      double ne[SFO_PNP_NSUM];
      sfo_pnp_normal_eq(&pb, inl, q, t, ne, scratch);
      const double v = (double)sqrtf((float)ne[27] / (float)n_inl);
      out->variance = v;
      out->variance_ang = v;
    }
  }
done:
  free(obj); free(dst3); free(dpx); free(dpy); free(orig); free(mask); free(mask_b); free(has3); free(e1); free(e2);
  free(scratch); free(counts);
  return rc;
}
