#pragma once
// k_ba.hip -- two-view bundle adjustment of one pass's motion estimate, for one candidate pair per 256-thread
// workgroup (included by k_verify.hip; runs inside the RANSAC / PnP workgroup right behind the estimate, while its
// inlier mask and compacted correspondences are still in LDS).
//
// Replaces myRegistrationVis.cpp:1192-1370 of the reference: Optimizer::optimizeBA [upstream rtabmap OptimizerG2O on
// g2o's sba types, un-vendored] on two poses (the "from" pose fixed at identity) and the inlier words' 3D points, mono
// or stereo reprojection residuals against both frames' keypoints, Huber kernel, Levenberg-Marquardt; words whose
// reprojection stays outside the kernel leave the inliers (:1314-1330), fewer than min_inliers left -> null
// (:1331-1336).  The algorithm and its operation order are fixed in DESIGN.md sections 3 and 4 (the test suite holds a
// CPU restatement of the same specification and compares results bit for bit).
//
// CDNA4 mapping: one LANE per inlier word (stride 256).  A word's two edges give its 3x3 point block, its 6x3 coupling
// to the pose and its share of the 6x6 pose block, all in registers; the point is eliminated in the lane (Schur
// complement, 3x3 inverse by cofactors) and only the 28 reduced sums (21 + 6 + cost) cross lanes, through the
// canonical block reduction (sfd::block_sum_canon_to_lds: totals land in LDS, every lane then solves the same 6x6
// system).  The back-substitution recomputes the word's blocks instead of keeping 27 doubles per word alive.
// ~60 fp64 values are live per lane, so the kernels that include this body are built for 2 workgroups per CU.
#include "sf_device_math.hpp"
#include "sf_internal.hpp"
#include "sf_pnp_math.hpp"

namespace {

#define BA_NSUM 28

struct BaLds {
  double* X;      // [kcap][3] current points (world = "from" base frame)
  double* Xc;     // [kcap][3] candidate points
  float* o1;      // [kcap][3] camera-1 observation: u - cx, v - cy, depth (<= 0: mono)
  float* o2;      // [kcap][3] camera-2 observation
  double* red;    // [4][32]
  double* ne_a;   // [32]
  double* ne_b;   // [32]
  int* misc;      // [16]
};

struct BaCam {
  double fx, fy, b, info, delta;
  double R1[9], t1[3];
};

// one edge: residual rows (2 mono, 3 stereo) and their Jacobians wrt the camera-frame point; 0 = behind the camera
__device__ __forceinline__ int ba_edge(const BaCam& cam, const double (&P)[3], const float* o, double (&r)[3],
                                       double (&J)[3][3]) {
  if (!(P[2] > 0.0)) return 0;
  const double iz = 1.0 / P[2];
  const double xn = P[0] * iz, yn = P[1] * iz;
  const double a0 = cam.fx * iz, a2 = -(a0 * xn);
  const double b1 = cam.fy * iz, b2 = -(b1 * yn);
  r[0] = cam.fx * xn - (double)o[0];
  r[1] = cam.fy * yn - (double)o[1];
  J[0][0] = a0; J[0][1] = 0.0; J[0][2] = a2;
  J[1][0] = 0.0; J[1][1] = b1; J[1][2] = b2;
  if (cam.b > 0.0 && o[2] > 0.0f) {
    const double fb = cam.fx * cam.b;
    r[2] = (cam.fx * xn - fb * iz) - ((double)o[0] - fb / (double)o[2]);
    J[2][0] = a0; J[2][1] = 0.0; J[2][2] = a2 + (fb * iz) * iz;
    return 3;
  }
  return 2;
}

__device__ __forceinline__ double ba_huber(double chi2, double delta, double& w) {
  const double e = sqrt(chi2);
  if (e <= delta) { w = 1.0; return chi2; }
  w = delta / e;
  return 2.0 * delta * e - delta * delta;
}

struct BaBlocks {
  double A[6], bp[3], C[18], H[21], bc[6], cost;
  bool ok;
};

__device__ inline void ba_point_blocks(const BaCam& cam, const double* Xp, const float* o1, const float* o2,
                                       const double (&R2)[9], const double (&t2)[3], BaBlocks& B) {
#pragma unroll
  for (int k = 0; k < 6; ++k) B.A[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) B.bp[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 18; ++k) B.C[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 21; ++k) B.H[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) B.bc[k] = 0.0;
  B.cost = 0.0;
  B.ok = true;
  const double X[3] = {Xp[0], Xp[1], Xp[2]};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = c == 0 ? cam.R1[k] : R2[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = c == 0 ? cam.t1[k] : t2[k];
    const float* o = c == 0 ? o1 : o2;
    const double Y[3] = {(R[0] * X[0] + R[1] * X[1]) + R[2] * X[2], (R[3] * X[0] + R[4] * X[1]) + R[5] * X[2],
                         (R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]};
    const double P[3] = {Y[0] + t[0], Y[1] + t[1], Y[2] + t[2]};
    double r[3], J[3][3];
    const int rows = ba_edge(cam, P, o, r, J);
    if (rows == 0) { B.cost += 1e30; B.ok = false; continue; }
    double chi2 = r[0] * r[0] + r[1] * r[1];
    if (rows == 3) chi2 = chi2 + r[2] * r[2];
    chi2 = chi2 * cam.info;
    double w;
    B.cost += ba_huber(chi2, cam.delta, w);
    w = w * cam.info;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < rows) {
        const double j0 = J[k][0], j1 = J[k][1], j2 = J[k][2];
        const double JX[3] = {(j0 * R[0] + j1 * R[3]) + j2 * R[6], (j0 * R[1] + j1 * R[4]) + j2 * R[7],
                              (j0 * R[2] + j1 * R[5]) + j2 * R[8]};
        const double wr = w * r[k];
        B.A[0] += w * (JX[0] * JX[0]); B.A[1] += w * (JX[0] * JX[1]); B.A[2] += w * (JX[0] * JX[2]);
        B.A[3] += w * (JX[1] * JX[1]); B.A[4] += w * (JX[1] * JX[2]); B.A[5] += w * (JX[2] * JX[2]);
        B.bp[0] += JX[0] * wr; B.bp[1] += JX[1] * wr; B.bp[2] += JX[2] * wr;
        if (c == 1) {
          const double Jc[6] = {j2 * Y[1] - j1 * Y[2], j0 * Y[2] - j2 * Y[0], j1 * Y[0] - j0 * Y[1], j0, j1, j2};
          int o_ = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int cc = a; cc < 6; ++cc) { B.H[o_] += w * (Jc[a] * Jc[cc]); ++o_; }
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            B.bc[a] += Jc[a] * wr;
            B.C[3 * a] += w * (Jc[a] * JX[0]); B.C[3 * a + 1] += w * (Jc[a] * JX[1]); B.C[3 * a + 2] += w * (Jc[a] * JX[2]);
          }
        }
      }
    }
  }
}

__device__ __forceinline__ bool ba_inv3(const double (&A)[6], double lambda, double (&Ai)[6]) {
  const double s = 1.0 + lambda;
  const double a = A[0] * s, b = A[1], c = A[2], d = A[3] * s, e = A[4], f = A[5] * s;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = (a * c00 + b * c01) + c * c02;
  if (!(det > 0.0) || !isfinite(det)) return false;
  const double id = 1.0 / det;
  Ai[0] = c00 * id; Ai[1] = c01 * id; Ai[2] = c02 * id;
  Ai[3] = (a * f - c * c) * id; Ai[4] = (b * c - a * e) * id; Ai[5] = (a * d - b * b) * id;
  return true;
}

__device__ __forceinline__ void ba_sym3_mul(const double (&Ai)[6], const double* v, double* out) {
  out[0] = (Ai[0] * v[0] + Ai[1] * v[1]) + Ai[2] * v[2];
  out[1] = (Ai[1] * v[0] + Ai[3] * v[1]) + Ai[4] * v[2];
  out[2] = (Ai[2] * v[0] + Ai[4] * v[1]) + Ai[5] * v[2];
}

// Schur-reduced normal equations at (q, t, points Xs): totals in `out` (LDS, [28])
// SMALL: at most 256 words (k_ba_pass's first launch): one word per lane of the canonical 256-lane scheme
template <int NW, bool SMALL = false>
__device__ inline void ba_normal_eq(const BaLds& L, const BaCam& cam, int m, const double* Xs, const double (&q)[4],
                                    const double (&t)[3], double lambda, double* out, int tid) {
  double R2[9];
  sfd::quat_to_R(q, R2);
  sfd::canon_reduce_to_lds<BA_NSUM, 32, NW, SMALL>(m, tid, L.red, out, [&](int i, double (&ne)[BA_NSUM]) {
    BaBlocks B;
    ba_point_blocks(cam, Xs + 3 * i, L.o1 + 3 * i, L.o2 + 3 * i, R2, t, B);
    double term[BA_NSUM];
#pragma unroll
    for (int k = 0; k < 21; ++k) term[k] = B.H[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) term[21 + k] = B.bc[k];
    term[27] = B.cost;
    double Ai[6];
    if (B.ok && ba_inv3(B.A, lambda, Ai)) {
      double CA[18];
#pragma unroll
      for (int a = 0; a < 6; ++a) ba_sym3_mul(Ai, B.C + 3 * a, CA + 3 * a);
      int o_ = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) {
          term[o_] = term[o_] - ((CA[3 * a] * B.C[3 * c] + CA[3 * a + 1] * B.C[3 * c + 1]) + CA[3 * a + 2] * B.C[3 * c + 2]);
          ++o_;
        }
#pragma unroll
      for (int a = 0; a < 6; ++a)
        term[21 + a] = term[21 + a] - ((CA[3 * a] * B.bp[0] + CA[3 * a + 1] * B.bp[1]) + CA[3 * a + 2] * B.bp[2]);
    }
#pragma unroll
    for (int k = 0; k < BA_NSUM; ++k) ne[k] += term[k];
  });
}

// candidate points Xc = X - A'^-1 (bp + C^T dc) at the current state
template <int NW>
__device__ inline void ba_backsub(const BaLds& L, const BaCam& cam, int m, const double (&q)[4], const double (&t)[3],
                                  double lambda, const double (&dc)[6], int tid) {
  double R2[9];
  sfd::quat_to_R(q, R2);
  for (int i = tid; i < m; i += 64 * NW) {
    BaBlocks B;
    ba_point_blocks(cam, L.X + 3 * i, L.o1 + 3 * i, L.o2 + 3 * i, R2, t, B);
    double Ai[6];
    double dx[3] = {0.0, 0.0, 0.0};
    if (B.ok && ba_inv3(B.A, lambda, Ai)) {
      double v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double sacc = B.bp[c];
#pragma unroll
        for (int a = 0; a < 6; ++a) sacc = sacc + B.C[3 * a + c] * dc[a];
        v[c] = sacc;
      }
      ba_sym3_mul(Ai, v, dx);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) L.Xc[3 * i + c] = L.X[3 * i + c] - dx[c];
  }
}

__device__ __forceinline__ BaLds ba_carve(unsigned char* p, int kcap) {
  BaLds L;
  L.X = (double*)p; p += (size_t)kcap * 24;
  L.Xc = (double*)p; p += (size_t)kcap * 24;
  L.red = (double*)p; p += 128 * 8;
  L.ne_a = (double*)p; p += 32 * 8;
  L.ne_b = (double*)p; p += 32 * 8;
  L.o1 = (float*)p; p += (size_t)kcap * 12;
  L.o2 = (float*)p; p += (size_t)kcap * 12;
  L.misc = (int*)p;
  return L;
}

__device__ __forceinline__ void ba_cam_setup(const DeviceParams& P, BaCam& cam) {
  cam.fx = P.fx; cam.fy = P.fy; cam.b = (double)P.stereo_baseline;
  cam.info = 1.0 / (double)P.ba_pixel_variance;
  cam.delta = (double)P.ba_robust_kernel_delta;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) cam.R1[3 * i + j] = (double)P.L[4 * j + i];
    cam.t1[i] = -(((double)P.L[i] * (double)P.L[3] + (double)P.L[4 + i] * (double)P.L[7]) + (double)P.L[8 + i] * (double)P.L[11]);
  }
}

// one inlier word -> its slot k of the adjustment's working set: the point, both observations with their depths
__device__ __forceinline__ void ba_put_word(const BaLds& L, const BaCam& cam, int k, float ax, float ay, float az, uint32_t c,
                                            const float4* kF, const float4* kT, const float* xT, bool to3d, float cxf,
                                            float cyf) {
  const float4 k1 = kF[c & 0xFFFFu], k2 = kT[c >> 16];
  L.X[3 * k] = (double)ax; L.X[3 * k + 1] = (double)ay; L.X[3 * k + 2] = (double)az;
  const float d1 = (float)(((cam.R1[6] * (double)ax + cam.R1[7] * (double)ay) + cam.R1[8] * (double)az) + cam.t1[2]);
  L.o1[3 * k] = k1.x - cxf; L.o1[3 * k + 1] = k1.y - cyf; L.o1[3 * k + 2] = d1;
  float d2 = 0.0f;
  if (to3d) {
    const float* bq = xT + 3 * (size_t)(c >> 16);
    const float bx = bq[0], by = bq[1], bz = bq[2];
    if (sfd::finite3(bx, by, bz))
      d2 = (float)(((cam.R1[6] * (double)bx + cam.R1[7] * (double)by) + cam.R1[8] * (double)bz) + cam.t1[2]);
  }
  L.o2[3 * k] = k2.x - cxf; L.o2[3 * k + 1] = k2.y - cyf; L.o2[3 * k + 2] = d2;
}

// The adjustment itself over the n words already in L.X / L.o1 / L.o2 (correspondence order), from the estimate p0;
// thread 0 writes the adjusted state to `ps`.
template <int NW, bool SMALL = false>
__device__ __forceinline__ void ba_solve(const BaLds& L, const BaCam& cam, int n, const PassState& p0, PassState& ps,
                                         const DeviceParams& P) {
  constexpr int NT = 64 * NW;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // camera 2: M = (T L)^-1 from float matrices (rtabmap::Transform), then double
  double q[4], t[3];
  {
    float TL[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        TL[4 * i + j] = (p0.T[4 * i] * P.L[j] + p0.T[4 * i + 1] * P.L[4 + j]) + p0.T[4 * i + 2] * P.L[8 + j];
      TL[4 * i + 3] = ((p0.T[4 * i] * P.L[3] + p0.T[4 * i + 1] * P.L[7]) + p0.T[4 * i + 2] * P.L[11]) + p0.T[4 * i + 3];
    }
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)TL[4 * j + i];
      t[i] = -(((double)TL[i] * (double)TL[3] + (double)TL[4 + i] * (double)TL[7]) + (double)TL[8 + i] * (double)TL[11]);
    }
    sfd::R_to_quat(R, q);
  }

  // ---- Levenberg-Marquardt on the Schur complement (every lane runs the same scalar control flow) -----------------------
  double* cur = L.ne_a;
  double* cnd = L.ne_b;
  double lambda = 1e-3;
  ba_normal_eq<NW, SMALL>(L, cam, n, L.X, q, t, lambda, cur, tid);
  for (int iter = 0; iter < P.ba_iterations; ++iter) {
    double d[6];
    if (!sfd::solve6(cur, lambda, d)) {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      ba_normal_eq<NW, SMALL>(L, cam, n, L.X, q, t, lambda, cur, tid);
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
#pragma unroll
    for (int i = 0; i < 4; ++i) qc[i] = qc[i] * qn;
#pragma unroll
    for (int i = 0; i < 3; ++i) tc[i] = t[i] + d[3 + i];
    ba_backsub<NW>(L, cam, n, q, t, lambda, d, tid);
    const double lambda_acc = lambda * 0.1 < 1e-16 ? 1e-16 : lambda * 0.1;
    __syncthreads();                                   // candidate points written
    ba_normal_eq<NW, SMALL>(L, cam, n, L.Xc, qc, tc, lambda_acc, cnd, tid);
    const double dd = ((((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]) + d[4] * d[4]) + d[5] * d[5];
    const double tt = ((tc[0] * tc[0] + tc[1] * tc[1]) + tc[2] * tc[2]) + 1.0;
    if (cnd[27] < cur[27]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = qc[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = tc[i];
      for (int i = tid; i < 3 * n; i += NT) L.X[i] = L.Xc[i];
      double* sw = cur; cur = cnd; cnd = sw;
      lambda = lambda_acc;
      __syncthreads();
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      ba_normal_eq<NW, SMALL>(L, cam, n, L.X, q, t, lambda, cur, tid);
    }
    if (dd <= 1.4e-14 * tt) break;
  }

  // ---- outliers: a word any of whose edges ends with chi2 > delta^2 ------------------------------------------------------
  double R2[9];
  sfd::quat_to_R(q, R2);
  const double lim = cam.delta * cam.delta;
  int n_out = 0;
  for (int i = tid; i < n; i += NT) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double R[9], tt[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = c == 0 ? cam.R1[k] : R2[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) tt[k] = c == 0 ? cam.t1[k] : t[k];
      const double* Xi = L.X + 3 * i;
      const double Pc[3] = {((R[0] * Xi[0] + R[1] * Xi[1]) + R[2] * Xi[2]) + tt[0],
                            ((R[3] * Xi[0] + R[4] * Xi[1]) + R[5] * Xi[2]) + tt[1],
                            ((R[6] * Xi[0] + R[7] * Xi[1]) + R[8] * Xi[2]) + tt[2]};
      double r[3], J[3][3];
      const int rows = ba_edge(cam, Pc, (c == 0 ? L.o1 : L.o2) + 3 * i, r, J);
      if (rows == 0) { bad = true; continue; }
      double chi2 = r[0] * r[0] + r[1] * r[1];
      if (rows == 3) chi2 = chi2 + r[2] * r[2];
      chi2 = chi2 * cam.info;
      if (chi2 > lim) bad = true;
    }
    n_out += bad ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n_out += __shfl_xor(n_out, off);
  __syncthreads();
  if (lane == 0) L.misc[8 + wave] = n_out;
  __syncthreads();
  n_out = L.misc[8];
#pragma unroll
  for (int w = 1; w < NW; ++w) n_out += L.misc[8 + w];

  if (tid == 0) {
    PassState o = p0;
    o.inliers = n - n_out;
    if (o.inliers < P.min_inliers) {
      o.is_null = 1;
#pragma unroll
      for (int i = 0; i < 12; ++i) o.T[i] = 0.f;
    } else {
      double Rd[9];
      sfd::quat_to_R(q, Rd);
      float Rf[9], tf[3], MR[9], Mt[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rf[i] = (float)Rd[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) tf[i] = (float)t[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          MR[3 * i + j] = (P.L[4 * i] * Rf[j] + P.L[4 * i + 1] * Rf[3 + j]) + P.L[4 * i + 2] * Rf[6 + j];
        Mt[i] = ((P.L[4 * i] * tf[0] + P.L[4 * i + 1] * tf[1]) + P.L[4 * i + 2] * tf[2]) + P.L[4 * i + 3];
      }
      bool allz = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o.T[4 * i + j] = MR[3 * j + i];
        o.T[4 * i + 3] = -((MR[i] * Mt[0] + MR[3 + i] * Mt[1]) + MR[6 + i] * Mt[2]);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) allz = allz && (o.T[i] == 0.f);
      o.is_null = allz ? 1 : 0;
    }
    ps = o;
  }
}

// Bundle adjustment of one pass.  pts[i] / cidx[i] / inl[i] (LDS, i < m): the estimate's compacted correspondences --
// the "from" 3D point, the packed (from | to << 16) feature indices, the inlier flag.  `ps` (the pass state the
// estimate just wrote; thread 0's view is authoritative): T, inliers and is_null are updated in place.
template <int NW = 4>
__device__ __forceinline__ void ba_body(const StoreView& st, int sF, int sT, const float4* pts, const uint32_t* cidx,
                                        const uint8_t* inl, int m, PassState& ps, const DeviceParams& P,
                                        unsigned char* lds) {
  constexpr int NT = 64 * NW;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const BaLds L = ba_carve(lds, kcap);
  __syncthreads();                       // the estimate's result (ps) is visible
  const PassState p0 = ps;
  // :1192-1197 gate (the words3From / wordsTo conditions hold whenever the estimate ran)
  if (p0.is_null || p0.inliers <= 0) return;

  BaCam cam;
  ba_cam_setup(P, cam);

  // ---- the inlier words, in correspondence order -----------------------------------------------------------------------
  const float4* kF = st.kp + (size_t)sF * kcap;
  const float4* kT = st.kp + (size_t)sT * kcap;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  const bool to3d = st.meta[sT].y > 0;
  const float cxf = (float)P.cx, cyf = (float)P.cy;
  if (tid < 16) L.misc[tid] = 0;
  __syncthreads();
  int n = 0;
  for (int base = 0; base < m; base += NT) {
    const int i = base + tid;
    const bool in = i < m && inl[i] != 0;
    const unsigned long long bal = __ballot(in);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) L.misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int c = L.misc[4 + w];
      if (w < wave) woff += c;
      total += c;
    }
    if (in) {
      const float4 a = pts[i];
      ba_put_word(L, cam, n + woff + before, a.x, a.y, a.z, cidx[i], kF, kT, xT, to3d, cxf, cyf);
    }
    n += total;
    __syncthreads();
  }
  if (n == 0) return;
  ba_solve<NW>(L, cam, n, p0, ps, P);
}

// The adjustment of one pass as a launch of its own (round 5): the estimate's kernel left its inlier set as one byte
// per "from" feature (`mask`, [kcap] of this pair) next to the pass's correspondence list, and the words are rebuilt
// from those -- the correspondences whose "from" point is finite (PnP) / whose two points are finite and non-zero
// (3D-3D), i.e. the estimate's own compaction, in list order, restricted to the inliers: the sequence ba_body takes
// from the estimate's LDS.  Same arithmetic, same bytes; what it buys is that the estimators are no longer compiled
// for the adjustment's ~60 live fp64 values (256 registers + 700 B of scratch per lane at 2 workgroups per CU in
// rounds 2-4) and the adjustment runs at the width its ~100 words fill.
// `cap`: words the workgroup's LDS holds (sf_ba_lds_bytes(cap)); the caller has checked the estimate's inlier count
// (an upper bound of the words) against it, and the gate of :1192-1197 (a non-null estimate with inliers).
// SMALL: cap <= 256.
template <int NW, bool PNP, bool SMALL>
__device__ __forceinline__ void ba_pass_body(const StoreView& st, int sF, int sT, const uint32_t* __restrict__ cl,
                                             int n_corr, const uint8_t* __restrict__ mask, const PassState& p0,
                                             PassState& ps, const DeviceParams& P, unsigned char* lds, int cap) {
  constexpr int NT = 64 * NW;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const BaLds L = ba_carve(lds, cap);
  BaCam cam;
  ba_cam_setup(P, cam);
  const float4* kF = st.kp + (size_t)sF * kcap;
  const float4* kT = st.kp + (size_t)sT * kcap;
  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  const bool to3d = st.meta[sT].y > 0;
  const float cxf = (float)P.cx, cyf = (float)P.cy;
  if (tid < 16) L.misc[tid] = 0;
  __syncthreads();
  int n = 0;
  for (int base = 0; base < n_corr; base += NT) {
    const int i = base + tid;
    bool in = false;
    uint32_t c = 0;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < n_corr) {
      c = cl[i];
      if (mask[c & 0xFFFFu]) {
        const float* a = xF + 3 * (c & 0xFFFFu);
        ax = a[0]; ay = a[1]; az = a[2];
        in = sfd::finite3(ax, ay, az);
        if (!PNP && in) {      // util3d::findCorrespondences: both points finite and non-zero
          const float* b = xT + 3 * (size_t)(c >> 16);
          const float bx = b[0], by = b[1], bz = b[2];
          in = sfd::finite3(bx, by, bz) && (ax != 0.f || ay != 0.f || az != 0.f) && (bx != 0.f || by != 0.f || bz != 0.f);
        }
      }
    }
    const unsigned long long bal = __ballot(in);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) L.misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int cw = L.misc[4 + w];
      if (w < wave) woff += cw;
      total += cw;
    }
    if (in && n + woff + before < cap) ba_put_word(L, cam, n + woff + before, ax, ay, az, c, kF, kT, xT, to3d, cxf, cyf);
    n += total;
    __syncthreads();
  }
  if (n == 0 || n > cap) return;      // (n <= the estimate's inlier count <= cap by the caller's check)
  ba_solve<NW, SMALL>(L, cam, n, p0, ps, P);
}

// Vis/ForwardEstOnly = false WITH bundle adjustment (myRegistrationVis.cpp:1155-1197, :1369, :1376-1394): the two
// directions' estimates of a pass (stage kernels k_ransac<.,0/1> / k_pnp<.,0/1>, which left their inlier masks -- one
// byte per "from" feature -- in HBM) are merged here.  The adjustment refines the FORWARD transform over the UNION of the
// inliers and ends with transforms[1].setNull(): the backward estimate then only contributed inliers.  Where its gate is
// closed (no forward transform, no inlier) the directions merge as without it (inverse of the backward transform,
// interpolate(0.5), mean covariance).  Words of the union without a finite point in the "from" frame (inliers of the
// backward PnP only) stay out of the adjustment and of the inlier count behind it.
// LDS: [kcap] float4 points | [kcap] u32 packed indices | [kcap] u8 ones | 16 ints, then ba_body's working set.
__host__ __device__ inline size_t sf_merge_ba_head_bytes(int kcap) {
  return (((size_t)kcap * 16 + (size_t)kcap * 4 + (size_t)kcap + 16 * 4) + 15) & ~(size_t)15;
}

template <bool PNP>
__global__ void __launch_bounds__(SF_BLOCK, 2)
k_merge_directions_ba(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
                      const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
                      const uint32_t* __restrict__ corr, const CorrHeader* __restrict__ hdr,
                      const uint8_t* __restrict__ guided_flag, PassState* __restrict__ fwd,
                      const PassState* __restrict__ back, const uint8_t* __restrict__ mask_f,
                      const uint8_t* __restrict__ mask_b, int extra_3dof, DeviceParams P) {
  if ((int)blockIdx.x >= *counter) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = list[blockIdx.x], tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  float4* pts = (float4*)smem_raw;
  uint32_t* cidx = (uint32_t*)(smem_raw + (size_t)kcap * 16);
  uint8_t* ones = smem_raw + (size_t)kcap * 20;
  int* misc = (int*)(smem_raw + ((((size_t)kcap * 21) + 3) & ~(size_t)3));
  const CorrHeader h = hdr[pair];
  const bool guided = guided_flag != nullptr && guided_flag[pair] != 0;
  const bool g0 = !PNP || sf_pnp_dir_gate(0, h, st.meta[sF].x, guided, P.min_inliers);
  const bool g1 = !PNP || sf_pnp_dir_gate(1, h, st.meta[sF].x, guided, P.min_inliers);
  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  const uint32_t* cl = corr + (size_t)pair * kcap;
  const uint8_t* mf = mask_f + (size_t)pair * kcap;
  const uint8_t* mb = mask_b + (size_t)pair * kcap;
  if (tid < 16) misc[tid] = 0;
  __syncthreads();
  // the union of the inliers (all of it) and, in correspondence order, its words with a finite "from" point
  int uni = 0, uni_m = 0, n_ba = 0;
  for (int base = 0; base < h.n_corr; base += SF_BLOCK) {
    const int i = base + tid;
    bool in = false, ok = false;
    uint32_t c = 0;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < h.n_corr) {
      c = cl[i];
      in = (mf[c & 0xFFFFu] | mb[c & 0xFFFFu]) != 0;
      const float* a = xF + 3 * (c & 0xFFFFu);
      ax = a[0]; ay = a[1]; az = a[2];
      const bool fa = sfd::finite3(ax, ay, az);
      ok = in && fa;
      if (PNP) {
        bool m = g0 && fa;
        if (g1 && !m) { const float* b = xT + 3 * (c >> 16); m = sfd::finite3(b[0], b[1], b[2]); }
        uni_m += m ? 1 : 0;
      }
    }
    uni += in ? 1 : 0;
    const unsigned long long bal = __ballot(ok);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SF_BLOCK / 64; ++w) {
      const int cw = misc[4 + w];
      if (w < wave) woff += cw;
      total += cw;
    }
    if (ok) {
      const int k = n_ba + woff + before;
      pts[k] = make_float4(ax, ay, az, 0.f);
      cidx[k] = c;
      ones[k] = 1;
    }
    n_ba += total;
    __syncthreads();
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { uni += __shfl_xor(uni, off); uni_m += __shfl_xor(uni_m, off); }
  if (lane == 0) { atomicAdd(&misc[0], uni); atomicAdd(&misc[1], uni_m); }
  __syncthreads();
  uni = misc[0];
  uni_m = misc[1];
  if (tid == 0) {
    PassState a = fwd[pair];
    const PassState b = back[pair];
    a.matches = PNP ? uni_m : (a.matches > b.matches ? a.matches : b.matches);
    // :1192-1197 (the words3From / wordsTo conditions hold whenever a forward transform exists)
    const bool adjust = P.bundle_adjustment != 0 && !a.is_null && n_ba > 0;
    misc[2] = adjust ? 1 : 0;
    if (adjust) {
      a.inliers = n_ba;
    } else {
      a.inliers = uni;
      if (!b.is_null) {
        float inv[12];
        sfd::rigid_inverse_canon(b.T, inv);
        if (a.is_null) {
#pragma unroll
          for (int i = 0; i < 12; ++i) a.T[i] = inv[i];
          a.is_null = 0;
          a.var = b.var;
          a.var_ang = b.var_ang;
        } else {
          float mid[12];
          sfd::interpolate_half_canon(a.T, inv, mid);
#pragma unroll
          for (int i = 0; i < 12; ++i) a.T[i] = mid[i];
          a.var = (a.var + b.var) / 2.0;
          a.var_ang = (a.var_ang + b.var_ang) / 2.0;
        }
      }
    }
    fwd[pair] = a;
  }
  __syncthreads();
  if (misc[2]) ba_body(st, sF, sT, pts, cidx, ones, n_ba, fwd[pair], P, smem_raw + sf_merge_ba_head_bytes(kcap));
  if (extra_3dof && tid == 0 && !fwd[pair].is_null)
    for (int t = 0; t < extra_3dof; ++t) sfd::to3dof_canon(fwd[pair].T);
}

}  // namespace

size_t sf_ba_lds_bytes(int kcap) {
  return (size_t)kcap * (24 + 24 + 12 + 12) + 128 * 8 + 2 * 32 * 8 + 16 * 4;
}

// (both launchers of the stage pipeline end here when both options are on)
int sf_launch_merge_directions_ba(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass,
                                  bool pnp, const uint8_t* mask_f, const uint8_t* mask_b) {
  const size_t lds = sf_merge_ba_head_bytes(st.kcap) + sf_ba_lds_bytes(st.kcap);
  if (lds > 160 * 1024) return sf_fail(c, SF_ERANGE, "bundle adjustment of both directions needs %zu B of LDS (> 160 KiB)", lds);
  if (!c->merge_ba_attr_set) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_merge_directions_ba<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SF_HIP(c, hipFuncSetAttribute((const void*)k_merge_directions_ba<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->merge_ba_attr_set = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  const int32_t* list = (const int32_t*)(pass == 1 ? c->list1.p : c->list3.p);
  const int32_t* counter = counters + (pass == 1 ? 0 : 2);
  const uint32_t* corr = (const uint32_t*)(pass == 1 ? c->corr1.p : c->corr2.p);
  const CorrHeader* hdr = (const CorrHeader*)(pass == 1 ? c->hdr1.p : c->hdr2.p);
  PassState* ps = (PassState*)(pass == 1 ? c->pass1.p : c->pass2.p);
  const int end_3dof = c->dparams.force_3dof ? (pass == 1 ? 2 : 1) : 0;
  const uint8_t* guided_flag = pass == 2 ? (const uint8_t*)c->flags.p : nullptr;
  if (pnp)
    hipLaunchKernelGGL(k_merge_directions_ba<true>, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to, list, counter,
                       corr, hdr, guided_flag, ps, (const PassState*)c->pass_back.p, mask_f, mask_b, end_3dof, c->dparams);
  else
    hipLaunchKernelGGL(k_merge_directions_ba<false>, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to, list, counter,
                       corr, hdr, guided_flag, ps, (const PassState*)c->pass_back.p, mask_f, mask_b, end_3dof, c->dparams);
  return SF_OK;
}
