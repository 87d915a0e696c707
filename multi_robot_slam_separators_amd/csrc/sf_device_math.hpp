// sf_device_math.hpp -- device-side numerics of the RANSAC / guided-matching kernels.
//
// DESIGN.md "Canonical arithmetic" fixes the ORDER of every floating-point operation so that the
// GPU and the CPU restatement can be compared bit for bit.  This file is the GPU statement of
// that order; translation units including it are compiled with -ffp-contract=off and use
// explicit fma where the canonical form has one.
//
// Upstream algorithms restated (none of this code exists in the reference repository; it calls
// un-vendored rtabmap / PCL):
//   rigid fit        pcl::SampleConsensusModelRegistration::estimateRigidTransformationSVD
//                    (pcl::umeyama, double) -- solved here with Horn's unit quaternion
//   sample test      pcl::SampleConsensusModelRegistration::isSampleGood
//   adaptive k       pcl::RandomSampleConsensus::computeModel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfd {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Three distinct indices in [0,m): stateless stand-in for PCL's drawIndexSample.
__device__ __forceinline__ void sample_triplet(uint64_t seed, uint32_t it, uint32_t attempt, uint32_t m,
                                               uint32_t& i0, uint32_t& i1, uint32_t& i2) {
  uint64_t ha = mix64(seed ^ mix64(((uint64_t)it << 32) | (uint64_t)attempt));
  uint64_t hb = mix64(ha);
  uint32_t r0 = (uint32_t)(ha >> 32), r1 = (uint32_t)ha, r2 = (uint32_t)(hb >> 32);
  i0 = __umulhi(r0, m);
  i1 = __umulhi(r1, m - 1);
  if (i1 >= i0) ++i1;
  i2 = __umulhi(r2, m - 2);
  uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
  if (i2 >= lo) ++i2;
  if (i2 >= hi) ++i2;
}

// ln(x) from IEEE + - * / only (bit-identical to the CPU restatement's series).
__device__ inline double canon_log(double x) {
  int e;
  double m = frexp(x, &e);
  if (m < 0.70710678118654752440) { m = m * 2.0; e -= 1; }
  double z = (m - 1.0) / (m + 1.0);
  double z2 = z * z;
  double s = 1.0 / 27.0;
#pragma unroll
  for (int k = 12; k >= 0; --k) s = s * z2 + 1.0 / (double)(2 * k + 1);
  return 2.0 * z * s + (double)e * 0.69314718055994530942;
}

// Cyclic Jacobi on a symmetric N x N matrix held in registers (all indices compile-time).
template <int N>
__device__ inline void jacobi(double (&a)[N][N], double (&v)[N][N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 50; ++sweep) {
    double sm = 0.0;
#pragma unroll
    for (int p = 0; p < N - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < N; ++q) sm += fabs(a[p][q]);
    if (sm == 0.0) break;
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        double apq = a[p][q];
        double g = 100.0 * fabs(apq);
        double app = fabs(a[p][p]), aqq = fabs(a[q][q]);
        if (sweep > 3 && app + g == app && aqq + g == aqq) {
          a[p][q] = 0.0;
          a[q][p] = 0.0;
        } else if (apq != 0.0) {
          double h = a[q][q] - a[p][p];
          double t;
          if (fabs(h) + g == fabs(h)) {
            t = apq / h;
          } else {
            double theta = 0.5 * h / apq;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          double c = 1.0 / sqrt(1.0 + t * t);
          double s = t * c;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            double akp = a[k][p], akq = a[k][q];
            a[k][p] = c * akp - s * akq;
            a[k][q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < N; ++k) {
            double apk = a[p][k], aqk = a[q][k];
            a[p][k] = c * apk - s * aqk;
            a[q][k] = s * apk + c * aqk;
          }
          a[p][q] = 0.0;
          a[q][p] = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            double vkp = v[k][p], vkq = v[k][q];
            v[k][p] = c * vkp - s * vkq;
            v[k][q] = s * vkp + c * vkq;
          }
        }
      }
    }
  }
}

__device__ __forceinline__ double det3(double a, double b, double c, double d, double e, double f, double g,
                                       double h, double i) {
  return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}

// Eigenvalues (descending, clamped at 0) of a symmetric PSD 3x3 matrix given by its upper triangle:
// Newton on the characteristic cubic from trace(C), then the deflated quadratic.
__device__ inline void sym3_eigenvalues(double c00, double c01, double c02, double c11, double c12, double c22,
                                        double (&ev)[3]) {
  const double c2 = (c00 + c11) + c22;
  const double c1 = ((c00 * c11 - c01 * c01) + (c00 * c22 - c02 * c02)) + (c11 * c22 - c12 * c12);
  const double c0 = det3(c00, c01, c02, c01, c11, c12, c02, c12, c22);
  double x = c2;
  for (int it = 0; it < 50; ++it) {
    const double pv = ((x - c2) * x + c1) * x - c0;
    const double dp = (3.0 * x - 2.0 * c2) * x + c1;
    if (dp == 0.0) break;
    const double xn = x - pv / dp;
    if (!(xn < x)) break;
    const double step = x - xn;
    x = xn;
    if (step <= 1e-14 * fabs(xn)) break;
  }
  const double l1 = x > 0.0 ? x : 0.0;
  const double s = c2 - l1;
  const double q = c1 - l1 * s;
  double disc = s * s - 4.0 * q;
  if (!(disc > 0.0)) disc = 0.0;
  const double r = sqrt(disc);
  double l2 = 0.5 * (s + r), l3 = 0.5 * (s - r);
  if (!(l2 > 0.0)) l2 = 0.0;
  if (!(l3 > 0.0)) l3 = 0.0;
  ev[0] = l1; ev[1] = l2; ev[2] = l3;
}

// Rotation and translation from the cross-covariance S[j][k] = sum a_j b_k of the demeaned
// source/target, their means and spreads ga = sum |a|^2, gb = sum |b|^2 (Horn's quaternion
// method).  The dominant eigenpair of the 4x4 matrix N comes from Newton's iteration on the
// characteristic quartic, started at the upper bound (ga+gb)/2 (monotone convergence to the largest
// root), and the best-conditioned column of adj(N - lambda I): ~0.5 kflop and a dozen divisions
// instead of a full Jacobi eigen-decomposition (3 div + 2 sqrt per rotation).  A power iteration is the
// fallback for a vanishing adjugate.  Output as the float coefficients PCL stores.
__device__ inline void rigid_from_moments(const double (&S)[3][3], const double (&mp)[3],
                                          const double (&mq)[3], double ga, double gb, float (&coef)[12]) {
  double Nm[4][4];
  Nm[0][0] = (S[0][0] + S[1][1]) + S[2][2];
  Nm[1][1] = (S[0][0] - S[1][1]) - S[2][2];
  Nm[2][2] = (S[1][1] - S[0][0]) - S[2][2];
  Nm[3][3] = (S[2][2] - S[0][0]) - S[1][1];
  Nm[0][1] = Nm[1][0] = S[1][2] - S[2][1];
  Nm[0][2] = Nm[2][0] = S[2][0] - S[0][2];
  Nm[0][3] = Nm[3][0] = S[0][1] - S[1][0];
  Nm[1][2] = Nm[2][1] = S[0][1] + S[1][0];
  Nm[1][3] = Nm[3][1] = S[2][0] + S[0][2];
  Nm[2][3] = Nm[3][2] = S[1][2] + S[2][1];

  double ss = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) ss += S[j][k] * S[j][k];
  const double c2 = -2.0 * ss;
  const double c1 = -8.0 * det3(S[0][0], S[0][1], S[0][2], S[1][0], S[1][1], S[1][2], S[2][0], S[2][1], S[2][2]);
  const double m0 = det3(Nm[1][1], Nm[1][2], Nm[1][3], Nm[2][1], Nm[2][2], Nm[2][3], Nm[3][1], Nm[3][2], Nm[3][3]);
  const double m1 = det3(Nm[1][0], Nm[1][2], Nm[1][3], Nm[2][0], Nm[2][2], Nm[2][3], Nm[3][0], Nm[3][2], Nm[3][3]);
  const double m2 = det3(Nm[1][0], Nm[1][1], Nm[1][3], Nm[2][0], Nm[2][1], Nm[2][3], Nm[3][0], Nm[3][1], Nm[3][3]);
  const double m3 = det3(Nm[1][0], Nm[1][1], Nm[1][2], Nm[2][0], Nm[2][1], Nm[2][2], Nm[3][0], Nm[3][1], Nm[3][2]);
  const double c0 = ((Nm[0][0] * m0 - Nm[0][1] * m1) + Nm[0][2] * m2) - Nm[0][3] * m3;

  double x = 0.5 * (ga + gb);
  for (int it = 0; it < 50; ++it) {
    const double x2 = x * x;
    const double b = (x2 + c2) * x;
    const double a = b + c1;
    const double pv = a * x + c0;
    const double dp = (2.0 * x2 * x + b) + a;
    if (dp == 0.0) break;
    const double xn = x - pv / dp;
    if (!(xn < x)) break;
    const double step = x - xn;
    x = xn;
    if (step <= 1e-14 * fabs(xn)) break;
  }

  const double b00 = Nm[0][0] - x, b11 = Nm[1][1] - x, b22 = Nm[2][2] - x, b33 = Nm[3][3] - x;
  const double b01 = Nm[0][1], b02 = Nm[0][2], b03 = Nm[0][3], b12 = Nm[1][2], b13 = Nm[1][3], b23 = Nm[2][3];
  const double a00 = det3(b11, b12, b13, b12, b22, b23, b13, b23, b33);
  const double a11 = det3(b00, b02, b03, b02, b22, b23, b03, b23, b33);
  const double a22 = det3(b00, b01, b03, b01, b11, b13, b03, b13, b33);
  const double a33 = det3(b00, b01, b02, b01, b11, b12, b02, b12, b22);
  const double a01 = -det3(b01, b12, b13, b02, b22, b23, b03, b23, b33);
  const double a02 = det3(b01, b11, b13, b02, b12, b23, b03, b13, b33);
  const double a03 = -det3(b01, b11, b12, b02, b12, b22, b03, b13, b23);
  const double a12 = -det3(b00, b01, b03, b02, b12, b23, b03, b13, b33);
  const double a13 = det3(b00, b01, b02, b02, b12, b22, b03, b13, b23);
  const double a23 = -det3(b00, b01, b02, b01, b11, b12, b03, b13, b23);
  double best = fabs(a00);
  double w = a00, qx = a01, qy = a02, qz = a03;
  if (fabs(a11) > best) { best = fabs(a11); w = a01; qx = a11; qy = a12; qz = a13; }
  if (fabs(a22) > best) { best = fabs(a22); w = a02; qx = a12; qy = a22; qz = a23; }
  if (fabs(a33) > best) { best = fabs(a33); w = a03; qx = a13; qy = a23; qz = a33; }
  double nrm2 = ((w * w + qx * qx) + qy * qy) + qz * qz;
  if (!(best > 0.0) || !(nrm2 > 0.0) || !isfinite(nrm2)) {
    // vanishing adjugate (an exactly degenerate configuration: the largest eigenvalue is repeated, or N = 0): a vector
    // of its eigenspace by 64 steps of the power iteration on N + shift I (positive semi-definite with
    // shift = (ga + gb) / 2 >= |eigenvalues|), from a fixed start.  Round 1 ran a cyclic Jacobi here: its two 4x4
    // double matrices raised the 3-point fit from 72 to 110 VGPRs and were what pushed k_ransac / k_verify_fused into
    // scratch (116-128 bytes per lane, 125 MB of spill writes per 10 000-pair launch) although this branch is
    // practically never taken.  Any unit vector of the eigenspace is a valid answer.
    const double shift = 0.5 * (ga + gb);
    double v0 = 1.0, v1 = 0.5, v2 = 0.25, v3 = 0.125;
    for (int it = 0; it < 64; ++it) {
      const double u0 = (((Nm[0][0] + shift) * v0 + Nm[0][1] * v1) + Nm[0][2] * v2) + Nm[0][3] * v3;
      const double u1 = ((Nm[0][1] * v0 + (Nm[1][1] + shift) * v1) + Nm[1][2] * v2) + Nm[1][3] * v3;
      const double u2 = ((Nm[0][2] * v0 + Nm[1][2] * v1) + (Nm[2][2] + shift) * v2) + Nm[2][3] * v3;
      const double u3 = ((Nm[0][3] * v0 + Nm[1][3] * v1) + Nm[2][3] * v2) + (Nm[3][3] + shift) * v3;
      const double n2 = ((u0 * u0 + u1 * u1) + u2 * u2) + u3 * u3;
      if (!(n2 > 0.0) || !isfinite(n2)) break;
      const double in = 1.0 / sqrt(n2);
      v0 = u0 * in; v1 = u1 * in; v2 = u2 * in; v3 = u3 * in;
    }
    w = v0; qx = v1; qy = v2; qz = v3;
    nrm2 = ((w * w + qx * qx) + qy * qy) + qz * qz;
  }
  const double inv = 1.0 / sqrt(nrm2);
  w = w * inv;
  const double x_ = qx * inv, y = qy * inv, z = qz * inv;
  const double xx = x_ * x_, yy = y * y, zz = z * z, xy = x_ * y, xz = x_ * z, yz = y * z;
  const double wx = w * x_, wy = w * y, wz = w * z;
  double R[9];
  R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double t = mq[j] - ((R[3 * j] * mp[0] + R[3 * j + 1] * mp[1]) + R[3 * j + 2] * mp[2]);
    coef[4 * j + 0] = (float)R[3 * j + 0];
    coef[4 * j + 1] = (float)R[3 * j + 1];
    coef[4 * j + 2] = (float)R[3 * j + 2];
    coef[4 * j + 3] = (float)t;
  }
}

// Squared residual of one correspondence under float coefficients (canonical fma chain).
__device__ __forceinline__ float residual2(const float (&c)[12], float px, float py, float pz,
                                           float qx, float qy, float qz) {
  float tx = __fmaf_rn(c[2], pz, __fmaf_rn(c[1], py, __fmaf_rn(c[0], px, c[3])));
  float ty = __fmaf_rn(c[6], pz, __fmaf_rn(c[5], py, __fmaf_rn(c[4], px, c[7])));
  float tz = __fmaf_rn(c[10], pz, __fmaf_rn(c[9], py, __fmaf_rn(c[8], px, c[11])));
  float dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}


// Block-wide sum of N doubles per thread (256 threads) in the canonical order of DESIGN.md section 4:
// inside each wavefront the xor butterfly over lane distances 32, 16, 8, 4, 2, 1 (own + partner), then
// the four wave totals folded left to right.  The butterfly is evaluated TRANSPOSED: at every stage a
// lane keeps half of the values it still holds and trades the other half with its partner, so the
// additions are exactly the butterfly's (a + b is commutative in IEEE arithmetic, both lanes of a pair
// would have computed the same sum) but only ~N cross-lane moves are issued instead of 6 N -- the
// full butterfly made the LDS crossbar the bottleneck of the refinement loops.
// red: [4][STRIDE] doubles of LDS scratch, STRIDE >= the power of two above N.
template <int C>
__device__ __forceinline__ void sum_stage(double (&w)[32], int& idx, int lane, int off) {
  if constexpr (C > 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < C / 2; ++k) {
      const double send = up ? w[k] : w[k + C / 2];
      const double keep = up ? w[k + C / 2] : w[k];
      w[k] = keep + __shfl_xor(send, off);
    }
    idx += up ? C / 2 : 0;
  } else {
    w[0] = w[0] + __shfl_xor(w[0], off);
  }
}

template <int N, int STRIDE>
__device__ __forceinline__ void block_sum_canon(double (&v)[N], double* red, int tid) {
  static_assert(N >= 1 && N <= 32 && STRIDE >= N, "at most 32 values");
  constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : 32;
  static_assert(STRIDE >= P, "scratch rows must hold the padded count");
  const int lane = tid & 63, wave = tid >> 6;
  double w[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) w[k] = k < N ? v[k] : 0.0;
  int idx = 0;
  sum_stage<P>(w, idx, lane, 32);
  sum_stage<(P / 2 > 1 ? P / 2 : 1)>(w, idx, lane, 16);
  sum_stage<(P / 4 > 1 ? P / 4 : 1)>(w, idx, lane, 8);
  sum_stage<(P / 8 > 1 ? P / 8 : 1)>(w, idx, lane, 4);
  sum_stage<(P / 16 > 1 ? P / 16 : 1)>(w, idx, lane, 2);
  sum_stage<(P / 32 > 1 ? P / 32 : 1)>(w, idx, lane, 1);
  __syncthreads();  // previous users of `red` are done
  red[wave * STRIDE + idx] = w[0];   // every lane of a group holds the same total: identical writes
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = ((red[k] + red[STRIDE + k]) + red[2 * STRIDE + k]) + red[3 * STRIDE + k];
}

// The same N sums computed by ONE or TWO wavefronts for all 256 lanes of the canonical scheme (motion-estimation
// chains on fewer wavefronts, k_verify.hip): `acc(i, v)` adds element i's terms to v[0..N); a wavefront plays
// 4 / NW wavefronts of the 256-thread version one after the other -- the same strided partials, the same in-wave
// stages, the same four-row fold, hence the same bits.  NW = 4 is the 256-thread form (partials +
// block_sum_canon).  tid = 64 * (wavefront index among the NW) + lane.  `red` as above.
template <int N, int STRIDE, int NW, class F>
__device__ __forceinline__ void canon_reduce(int m, int tid, double* red, double (&out)[N], F acc) {
  if constexpr (NW == 4) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = 0.0;
    for (int i = tid; i < m; i += 256) acc(i, out);
    block_sum_canon<N, STRIDE>(out, red, tid);
  } else {
    static_assert(NW == 1 || NW == 2, "one, two or four wavefronts");
    constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : 32;
    static_assert(STRIDE >= P, "scratch rows must hold the padded count");
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();  // previous users of `red` are done (only the live wavefronts take part)
#pragma unroll 1
    for (int j = 0; j < 4 / NW; ++j) {
      const int vw = wave * (4 / NW) + j;     // the wavefront of the 256-thread scheme this pass stands in for
      if (64 * vw >= m) {                     // no element for this wavefront: its partials are all +0.0, and so are their sums
        if (lane < P) red[vw * STRIDE + lane] = 0.0;
        continue;
      }
      double w[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w[k] = 0.0;
      {
        double v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = 0.0;
        for (int i = lane + 64 * vw; i < m; i += 256) acc(i, v);
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = v[k];
      }
      int idx = 0;
      sum_stage<P>(w, idx, lane, 32);
      sum_stage<(P / 2 > 1 ? P / 2 : 1)>(w, idx, lane, 16);
      sum_stage<(P / 4 > 1 ? P / 4 : 1)>(w, idx, lane, 8);
      sum_stage<(P / 8 > 1 ? P / 8 : 1)>(w, idx, lane, 4);
      sum_stage<(P / 16 > 1 ? P / 16 : 1)>(w, idx, lane, 2);
      sum_stage<(P / 32 > 1 ? P / 32 : 1)>(w, idx, lane, 1);
      red[vw * STRIDE + idx] = w[0];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = ((red[k] + red[STRIDE + k]) + red[2 * STRIDE + k]) + red[3 * STRIDE + k];
  }
}

// Same sums, but the N totals are left in LDS (out[0..N)) instead of in every thread's registers: for
// callers that only need them as operands of a short scalar computation (k_pnp's 28-value normal
// equations would otherwise pin 56 VGPRs per copy).  out must not alias red.
template <int N, int STRIDE>
__device__ __forceinline__ void block_sum_canon_to_lds(const double (&v)[N], double* red, double* out, int tid) {
  static_assert(N >= 1 && N <= 32 && STRIDE >= N, "at most 32 values");
  constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : 32;
  static_assert(STRIDE >= P, "scratch rows must hold the padded count");
  const int lane = tid & 63, wave = tid >> 6;
  double w[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) w[k] = k < N ? v[k] : 0.0;
  int idx = 0;
  sum_stage<P>(w, idx, lane, 32);
  sum_stage<(P / 2 > 1 ? P / 2 : 1)>(w, idx, lane, 16);
  sum_stage<(P / 4 > 1 ? P / 4 : 1)>(w, idx, lane, 8);
  sum_stage<(P / 8 > 1 ? P / 8 : 1)>(w, idx, lane, 4);
  sum_stage<(P / 16 > 1 ? P / 16 : 1)>(w, idx, lane, 2);
  sum_stage<(P / 32 > 1 ? P / 32 : 1)>(w, idx, lane, 1);
  __syncthreads();  // previous users of `red` / `out` are done
  red[wave * STRIDE + idx] = w[0];
  __syncthreads();
  if (tid < N) out[tid] = ((red[tid] + red[STRIDE + tid]) + red[2 * STRIDE + tid]) + red[3 * STRIDE + tid];
  __syncthreads();
}

// canon_reduce with the N totals left in LDS (out[0..N), out must not alias red): the 256-thread form accumulates
// per lane and calls block_sum_canon_to_lds; one or two wavefronts play the four of the canonical scheme one after
// the other (see canon_reduce).  `acc(i, v)` adds element i's terms to v[0..N).
// ONCE: the caller guarantees m <= 256 -- every lane of the canonical scheme has at most one element, so its partial is
// 0.0 + that element's terms and no accumulator stays live across a loop.
template <int N, int STRIDE, int NW, bool ONCE = false, class F>
__device__ __forceinline__ void canon_reduce_to_lds(int m, int tid, double* red, double* out, F acc) {
  if constexpr (NW == 4) {
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = 0.0;
    if constexpr (ONCE) { if (tid < m) acc(tid, v); }
    else { for (int i = tid; i < m; i += 256) acc(i, v); }
    block_sum_canon_to_lds<N, STRIDE>(v, red, out, tid);
  } else {
    static_assert(NW == 1 || NW == 2, "one, two or four wavefronts");
    constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : 32;
    static_assert(STRIDE >= P, "scratch rows must hold the padded count");
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();  // previous users of `red` / `out` are done
#pragma unroll 1
    for (int j = 0; j < 4 / NW; ++j) {
      const int vw = wave * (4 / NW) + j;
      if (64 * vw >= m) {                     // no element for this wavefront: its partials are all +0.0, and so are their sums
        if (lane < P) red[vw * STRIDE + lane] = 0.0;
        continue;
      }
      double w[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w[k] = 0.0;
      {
        double v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = 0.0;
        if constexpr (ONCE) { if (lane + 64 * vw < m) acc(lane + 64 * vw, v); }
        else { for (int i = lane + 64 * vw; i < m; i += 256) acc(i, v); }
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = v[k];
      }
      int idx = 0;
      sum_stage<P>(w, idx, lane, 32);
      sum_stage<(P / 2 > 1 ? P / 2 : 1)>(w, idx, lane, 16);
      sum_stage<(P / 4 > 1 ? P / 4 : 1)>(w, idx, lane, 8);
      sum_stage<(P / 8 > 1 ? P / 8 : 1)>(w, idx, lane, 4);
      sum_stage<(P / 16 > 1 ? P / 16 : 1)>(w, idx, lane, 2);
      sum_stage<(P / 32 > 1 ? P / 32 : 1)>(w, idx, lane, 1);
      red[vw * STRIDE + idx] = w[0];
    }
    __syncthreads();
    if (tid < N) out[tid] = ((red[tid] + red[STRIDE + tid]) + red[2 * STRIDE + tid]) + red[3 * STRIDE + tid];
    __syncthreads();
  }
}

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// ---- Transform::to3DoF / inverse / interpolate(0.5) [upstream rtabmap Transform.cpp] in canonical forms: fixed
// operation order in double on the float entries, -ffp-contract=off, no trigonometric call (the
// test suite's CPU restatement performs the same operations in the same order).
// to3DoF: Transform(x, y, 0, 0, 0, yaw), yaw = atan2(r21, r11) -> the rotation (r11, r21) / |(r11, r21)| about z.
__device__ inline void to3dof_canon(float* T) {
  const double r11 = (double)T[0], r21 = (double)T[4];
  const double h = sqrt(r11 * r11 + r21 * r21);
  float c = 1.0f, s = 0.0f;
  if (h > 0.0) { c = (float)(r11 / h); s = (float)(r21 / h); }
  const float x = T[3], y = T[7];
  T[0] = c; T[1] = -s; T[2] = 0.0f; T[3] = x;
  T[4] = s; T[5] = c; T[6] = 0.0f; T[7] = y;
  T[8] = 0.0f; T[9] = 0.0f; T[10] = 1.0f; T[11] = 0.0f;
}

__device__ inline void quat_from_rot_canon(const float* T, double (&q)[4]) {
  double m[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i][j] = (double)T[4 * i + j];
  const double tr = (m[0][0] + m[1][1]) + m[2][2];
  if (tr > 0.0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t; q[1] = (m[0][2] - m[2][0]) * t; q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(((m[i][i] - m[j][j]) - m[k][k]) + 1.0);
    double qv[3];
    qv[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    qv[j] = (m[j][i] + m[i][j]) * t;
    qv[k] = (m[k][i] + m[i][k]) * t;
    q[0] = qv[0]; q[1] = qv[1]; q[2] = qv[2];
  }
}

__device__ inline void rigid_inverse_canon(const float* T, float* out) {
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)T[4 * i + j];
    t[i] = (double)T[4 * i + 3];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) out[4 * i + j] = (float)R[3 * j + i];
    out[4 * i + 3] = (float)(-((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]));
  }
}

__device__ inline void interpolate_half_canon(const float* A, const float* B, float* out) {
  double qa[4], qb[4], q[4];
  quat_from_rot_canon(A, qa);
  quat_from_rot_canon(B, qb);
  const double d = ((qa[0] * qb[0] + qa[1] * qb[1]) + qa[2] * qb[2]) + qa[3] * qb[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = d < 0.0 ? qa[i] - qb[i] : qa[i] + qb[i];
  const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  out[0] = (float)(1.0 - (tyy + tzz)); out[1] = (float)(txy - twz);         out[2] = (float)(txz + twy);
  out[4] = (float)(txy + twz);         out[5] = (float)(1.0 - (txx + tzz)); out[6] = (float)(tyz - twx);
  out[8] = (float)(txz - twy);         out[9] = (float)(tyz + twx);         out[10] = (float)(1.0 - (txx + tyy));
#pragma unroll
  for (int i = 0; i < 3; ++i) out[4 * i + 3] = A[4 * i + 3] + 0.5f * (B[4 * i + 3] - A[4 * i + 3]);
}

}  // namespace sfd
