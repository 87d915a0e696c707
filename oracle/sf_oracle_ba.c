/*
 * sf_oracle_ba.c -- CPU ORACLE (test infrastructure, NOT product code; see sf_oracle.h) of the two-view bundle
 * adjustment that follows a pass's motion estimate in the reference:
 *
 *   ros_ws/src/multi_robot_separators/src/myRegistrationVis.cpp:1192-1370
 *     :1192-1197  gate: _bundleAdjustment > 0, estimation type < 2, forward transform not null, inliers, 3D words of
 *                 the "from" frame and 2D words of the "to" frame present
 *     :1205-1206  poses: 1 = identity ("from", FIXED: optimizeBA(rootId = 1, ...)), 2 = the estimated transform
 *     :1232-1270  camera models of both frames = the stereo model's left camera with Tx = -baseline * f
 *     :1274-1302  per inlier word: its 3D point in the "from" frame, its keypoint in "from" with the depth of that
 *                 point, its keypoint in "to" with the depth of the "to" 3D point (0 when there is none)
 *     :1305       sba->optimizeBA(1, poses, links, models, points3DMap, wordReferences, &sbaOutliers)  [upstream]
 *     :1314-1330  words in sbaOutliers leave allInliers; :1331-1336 fewer than _minInliers left -> null transform;
 *                 :1337-1340 else transforms[0] = the optimised pose 2
 *
 * [upstream, un-vendored and unpinned: rtabmap OptimizerG2O::optimizeBA on g2o's sba types]  What is restated here is
 * the published algorithm those types implement, with rtabmap's parameter names: cameras with fixed intrinsics,
 * pose 1 fixed, pose 2 and the 3D points free; per observation a mono (u, v) or, when the observation has a depth and
 * the baseline is positive, a stereo (u, v, u - disparity) reprojection residual with information 1 / pixel
 * variance; Huber kernel of width robust_kernel_delta on every edge (iteratively re-weighted); Levenberg-Marquardt
 * on the Schur complement (points eliminated), at most `ba_iterations` evaluations; afterwards a word any of whose
 * edges has chi2 > delta^2 is an outlier.  PARITY UNPINNED like the rest of the motion-estimation stage; the
 * arithmetic ORDER below is this repository's canonical one so that the HIP kernel
 * (csrc/k_ba.hip) can be compared bit for bit:
 *   * world = the "from" base frame; camera 1 = L^-1 (fixed), camera 2 = M = (T L)^-1, M parametrised by a unit
 *     quaternion and a translation, updated by a LEFT rotation perturbation (as the PnP refinement, sf_oracle_pnp.c);
 *   * per point: 3x3 point block A, 6x3 coupling C, the pose block, all in double; point block inverted by
 *     cofactors; S = Hcc - C A'^-1 C^T and g = bc - C A'^-1 bp summed over the points in BLOCK ORDER (28 sums);
 *   * LM: lambda from 1e-3, diagonals scaled by 1 + lambda, x 0.1 on an accepted step, x 10 otherwise.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sf_oracle.h"
#include "sf_oracle_internal.h"

#define BA_NSUM 28

typedef struct {
  int m;                /* inlier words */
  const double* X0;     /* unused after init */
  double* X;            /* [m][3] current points */
  const float* o1;      /* [m][3] camera-1 observation: u - cx, v - cy, depth (<= 0: mono) */
  const float* o2;      /* [m][3] camera-2 observation */
  double fx, fy, b;     /* b = baseline (0: mono only) */
  double R1[9], t1[3];  /* camera 1: world -> optical (L^-1), fixed */
  double info;          /* 1 / pixel variance */
  double delta;         /* Huber width (in sqrt(chi2) units) */
} ba_problem;

/* one edge: residual (2 or 3) and its Jacobian wrt the camera-frame point P (rows: a = (a0, 0, a2), b = (0, b1, b2),
 * s = (a0, 0, s2)); returns the number of residual rows, 0 when the point is not in front of the camera */
static int ba_edge(double fx, double fy, double b, const double P[3], const float* o, double r[3], double Ja[3],
                   double Jb[3], double Js[3]) {
  if (!(P[2] > 0.0)) return 0;
  const double iz = 1.0 / P[2];
  const double xn = P[0] * iz, yn = P[1] * iz;
  const double a0 = fx * iz, a2 = -(a0 * xn);
  const double b1 = fy * iz, b2 = -(b1 * yn);
  r[0] = fx * xn - (double)o[0];
  r[1] = fy * yn - (double)o[1];
  Ja[0] = a0; Ja[1] = 0.0; Ja[2] = a2;
  Jb[0] = 0.0; Jb[1] = b1; Jb[2] = b2;
  if (b > 0.0 && o[2] > 0.0f) {
    const double fb = fx * b;
    r[2] = (fx * xn - fb * iz) - ((double)o[0] - fb / (double)o[2]);
    Js[0] = a0; Js[1] = 0.0; Js[2] = a2 + (fb * iz) * iz;
    return 3;
  }
  return 2;
}

/* Huber weight of an edge with squared error chi2 (already multiplied by the information): returns rho, *w */
static double ba_huber(double chi2, double delta, double* w) {
  const double e = sqrt(chi2);
  if (e <= delta) { *w = 1.0; return chi2; }
  *w = delta / e;
  return 2.0 * delta * e - delta * delta;
}

typedef struct {
  double A[6];    /* point block, upper triangle xx xy xz yy yz zz */
  double bp[3];
  double C[18];   /* pose x point coupling, row-major [6][3] */
  double H[21];   /* pose block, upper triangle */
  double bc[6];
  double cost;
  int ok;
} ba_blocks;

static void ba_point_blocks(const ba_problem* pb, int i, const double R2[9], const double t2[3], ba_blocks* B) {
  memset(B, 0, sizeof(*B));
  B->ok = 1;
  const double* X = pb->X + 3 * i;
  for (int cam = 0; cam < 2; ++cam) {
    const double* R = cam == 0 ? pb->R1 : R2;
    const double* t = cam == 0 ? pb->t1 : t2;
    const float* o = (cam == 0 ? pb->o1 : pb->o2) + 3 * i;
    const double Y[3] = {(R[0] * X[0] + R[1] * X[1]) + R[2] * X[2], (R[3] * X[0] + R[4] * X[1]) + R[5] * X[2],
                         (R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]};
    const double P[3] = {Y[0] + t[0], Y[1] + t[1], Y[2] + t[2]};
    double r[3], J[3][3];
    const int rows = ba_edge(pb->fx, pb->fy, pb->b, P, o, r, J[0], J[1], J[2]);
    if (rows == 0) { B->cost += 1e30; B->ok = 0; continue; }   /* behind a camera: unacceptable state */
    double chi2 = r[0] * r[0] + r[1] * r[1];
    if (rows == 3) chi2 = chi2 + r[2] * r[2];
    chi2 = chi2 * pb->info;
    double w;
    B->cost += ba_huber(chi2, pb->delta, &w);
    w = w * pb->info;
    for (int k = 0; k < rows; ++k) {
      /* Jacobian row wrt the world point: JX = J[k] * R ; wrt the pose (camera 2): (-(J[k] x Y)... , J[k]) */
      const double* j = J[k];
      const double JX[3] = {(j[0] * R[0] + j[1] * R[3]) + j[2] * R[6], (j[0] * R[1] + j[1] * R[4]) + j[2] * R[7],
                            (j[0] * R[2] + j[1] * R[5]) + j[2] * R[8]};
      const double wr = w * r[k];
      B->A[0] += w * (JX[0] * JX[0]); B->A[1] += w * (JX[0] * JX[1]); B->A[2] += w * (JX[0] * JX[2]);
      B->A[3] += w * (JX[1] * JX[1]); B->A[4] += w * (JX[1] * JX[2]); B->A[5] += w * (JX[2] * JX[2]);
      B->bp[0] += JX[0] * wr; B->bp[1] += JX[1] * wr; B->bp[2] += JX[2] * wr;
      if (cam == 1) {
        /* dP/d(omega) = -[Y]x : row j -> (j x Y) with the sign of the PnP refinement's Jacobian */
        const double Jc[6] = {j[2] * Y[1] - j[1] * Y[2], j[0] * Y[2] - j[2] * Y[0], j[1] * Y[0] - j[0] * Y[1],
                              j[0], j[1], j[2]};
        int o_ = 0;
        for (int a = 0; a < 6; ++a)
          for (int c = a; c < 6; ++c) B->H[o_++] += w * (Jc[a] * Jc[c]);
        for (int a = 0; a < 6; ++a) {
          B->bc[a] += Jc[a] * wr;
          B->C[3 * a] += w * (Jc[a] * JX[0]); B->C[3 * a + 1] += w * (Jc[a] * JX[1]); B->C[3 * a + 2] += w * (Jc[a] * JX[2]);
        }
      }
    }
  }
}

/* inverse of the damped point block (diagonal x (1 + lambda)) by cofactors; 0 when singular */
static int ba_inv3(const double A[6], double lambda, double Ai[6]) {
  const double s = 1.0 + lambda;
  const double a = A[0] * s, b = A[1], c = A[2], d = A[3] * s, e = A[4], f = A[5] * s;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = (a * c00 + b * c01) + c * c02;
  if (!(det > 0.0) || !isfinite(det)) return 0;
  const double id = 1.0 / det;
  Ai[0] = c00 * id; Ai[1] = c01 * id; Ai[2] = c02 * id;
  Ai[3] = (a * f - c * c) * id; Ai[4] = (b * c - a * e) * id; Ai[5] = (a * d - b * b) * id;
  return 1;
}

static void ba_sym3_mul(const double Ai[6], const double v[3], double out[3]) {
  out[0] = (Ai[0] * v[0] + Ai[1] * v[1]) + Ai[2] * v[2];
  out[1] = (Ai[1] * v[0] + Ai[3] * v[1]) + Ai[4] * v[2];
  out[2] = (Ai[2] * v[0] + Ai[4] * v[1]) + Ai[5] * v[2];
}

/* Schur-reduced normal equations at (q, t, X): out[0..20] = S upper triangle, out[21..26] = g, out[27] = cost */
static void ba_normal_eq(const ba_problem* pb, const double q[4], const double t[3], double lambda,
                         double out[BA_NSUM], double* scratch /* 28 * m */) {
  double R2[9];
  sfo_quat_to_R(q, R2);
  const int m = pb->m;
  for (int i = 0; i < m; ++i) {
    ba_blocks B;
    ba_point_blocks(pb, i, R2, t, &B);
    double term[BA_NSUM];
    for (int k = 0; k < 21; ++k) term[k] = B.H[k];
    for (int k = 0; k < 6; ++k) term[21 + k] = B.bc[k];
    term[27] = B.cost;
    double Ai[6];
    if (B.ok && ba_inv3(B.A, lambda, Ai)) {
      double CA[18];   /* C A'^-1, [6][3] */
      for (int a = 0; a < 6; ++a) ba_sym3_mul(Ai, B.C + 3 * a, CA + 3 * a);
      int o_ = 0;
      for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) {
          term[o_] = term[o_] - ((CA[3 * a] * B.C[3 * c] + CA[3 * a + 1] * B.C[3 * c + 1]) + CA[3 * a + 2] * B.C[3 * c + 2]);
          ++o_;
        }
      for (int a = 0; a < 6; ++a)
        term[21 + a] = term[21 + a] - ((CA[3 * a] * B.bp[0] + CA[3 * a + 1] * B.bp[1]) + CA[3 * a + 2] * B.bp[2]);
    }
    for (int k = 0; k < BA_NSUM; ++k) scratch[(size_t)k * m + i] = term[k];
  }
  for (int k = 0; k < BA_NSUM; ++k) out[k] = sfo_block_sum(scratch + (size_t)k * m, m);
}

/* candidate points Xc = X - A'^-1 (bp + C^T dc) at the CURRENT state */
static void ba_backsub(const ba_problem* pb, const double q[4], const double t[3], double lambda, const double dc[6],
                       double* Xc) {
  double R2[9];
  sfo_quat_to_R(q, R2);
  for (int i = 0; i < pb->m; ++i) {
    ba_blocks B;
    ba_point_blocks(pb, i, R2, t, &B);
    double Ai[6];
    double dx[3] = {0.0, 0.0, 0.0};
    if (B.ok && ba_inv3(B.A, lambda, Ai)) {
      double v[3];
      for (int c = 0; c < 3; ++c) {
        double sacc = B.bp[c];
        for (int a = 0; a < 6; ++a) sacc = sacc + B.C[3 * a + c] * dc[a];
        v[c] = sacc;
      }
      ba_sym3_mul(Ai, v, dx);
    }
    for (int c = 0; c < 3; ++c) Xc[3 * i + c] = pb->X[3 * i + c] - dx[c];
  }
}

/* Two-view bundle adjustment of one pass.  corr / mask: the pass's correspondences and the motion estimate's inlier
 * mask over them.  T (in/out): p_from = T p_to.  Returns SF_OK; *n_inliers (in/out) loses the outlier words,
 * mask_out (optional) is the surviving inlier mask, *is_null set when fewer than min_inliers survive. */
int sfo_bundle_adjust(const sf_params* p, const float* xyz_from, const sf_keypoint* kp_from, const float* xyz_to,
                      const sf_keypoint* kp_to, const uint16_t* corr_from, const uint16_t* corr_to,
                      const uint8_t* mask, int n_corr, float T[12], int* n_inliers, int* is_null, uint8_t* mask_out) {
  int m = 0;
  for (int i = 0; i < n_corr; ++i) m += mask[i] ? 1 : 0;
  if (mask_out) memcpy(mask_out, mask, (size_t)(n_corr > 0 ? n_corr : 0));
  if (m == 0) return SF_OK;
  double* X = (double*)malloc((size_t)m * 3 * sizeof(double));
  double* Xc = (double*)malloc((size_t)m * 3 * sizeof(double));
  float* o1 = (float*)malloc((size_t)m * 3 * sizeof(float));
  float* o2 = (float*)malloc((size_t)m * 3 * sizeof(float));
  int* orig = (int*)malloc((size_t)m * sizeof(int));
  double* scratch = (double*)malloc((size_t)m * BA_NSUM * sizeof(double));
  if (!X || !Xc || !o1 || !o2 || !orig || !scratch) { free(X); free(Xc); free(o1); free(o2); free(orig); free(scratch); return SF_ENOMEM; }
  ba_problem pb;
  memset(&pb, 0, sizeof(pb));
  pb.fx = p->fx; pb.fy = p->fy; pb.b = (double)p->stereo_baseline;
  pb.info = 1.0 / (double)p->ba_pixel_variance;
  pb.delta = (double)p->ba_robust_kernel_delta;
  /* camera 1 = L^-1 (rigid inverse in double) */
  const float* L = p->local_transform;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) pb.R1[3 * i + j] = (double)L[4 * j + i];
    pb.t1[i] = -(((double)L[i] * (double)L[3] + (double)L[4 + i] * (double)L[7]) + (double)L[8 + i] * (double)L[11]);
  }
  const float cxf = (float)p->cx, cyf = (float)p->cy;
  int k = 0;
  for (int i = 0; i < n_corr; ++i) {
    if (!mask[i]) continue;
    const float* a = xyz_from + 3 * (size_t)corr_from[i];
    for (int c = 0; c < 3; ++c) X[3 * k + c] = (double)a[c];
    /* depth of the observation = z of the point in the optical frame (util3d::transformPoint(pt, invLocalTransform).z) */
    const float d1 = (float)(((pb.R1[6] * (double)a[0] + pb.R1[7] * (double)a[1]) + pb.R1[8] * (double)a[2]) + pb.t1[2]);
    o1[3 * k] = kp_from[corr_from[i]].x - cxf; o1[3 * k + 1] = kp_from[corr_from[i]].y - cyf; o1[3 * k + 2] = d1;
    float d2 = 0.0f;
    if (xyz_to) {
      const float* bq = xyz_to + 3 * (size_t)corr_to[i];
      if (sfo_finite3(bq))
        d2 = (float)(((pb.R1[6] * (double)bq[0] + pb.R1[7] * (double)bq[1]) + pb.R1[8] * (double)bq[2]) + pb.t1[2]);
    }
    o2[3 * k] = kp_to[corr_to[i]].x - cxf; o2[3 * k + 1] = kp_to[corr_to[i]].y - cyf; o2[3 * k + 2] = d2;
    orig[k] = i;
    ++k;
  }
  pb.m = m; pb.X = X; pb.o1 = o1; pb.o2 = o2;

  /* camera 2: M = (T L)^-1 = L^-1 T^-1, as float matrices like rtabmap::Transform, then double */
  double q[4], t[3];
  {
    float TL[12];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) TL[4 * i + j] = (T[4 * i] * L[j] + T[4 * i + 1] * L[4 + j]) + T[4 * i + 2] * L[8 + j];
      TL[4 * i + 3] = ((T[4 * i] * L[3] + T[4 * i + 1] * L[7]) + T[4 * i + 2] * L[11]) + T[4 * i + 3];
    }
    double R[9];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)TL[4 * j + i];
      t[i] = -(((double)TL[i] * (double)TL[3] + (double)TL[4 + i] * (double)TL[7]) + (double)TL[8 + i] * (double)TL[11]);
    }
    sfo_R_to_quat(R, q);
  }

  /* ---- Levenberg-Marquardt on the Schur complement ----------------------------------------------------------- */
  double ne[BA_NSUM], nc[BA_NSUM];
  double lambda = 1e-3;
  ba_normal_eq(&pb, q, t, lambda, ne, scratch);
  for (int iter = 0; iter < p->ba_iterations; ++iter) {
    double d[6];
    if (!sfo_pnp_solve6(ne, lambda, d)) {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      ba_normal_eq(&pb, q, t, lambda, ne, scratch);   /* the Schur complement depends on the damping */
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
    ba_backsub(&pb, q, t, lambda, d, Xc);
    /* evaluate the candidate (its normal equations with the damping an accepted step would continue with) */
    double* keep = pb.X;
    pb.X = Xc;
    const double lambda_acc = lambda * 0.1 < 1e-16 ? 1e-16 : lambda * 0.1;
    ba_normal_eq(&pb, qc, tc, lambda_acc, nc, scratch);
    pb.X = keep;
    const double dd = ((((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]) + d[4] * d[4]) + d[5] * d[5];
    const double tt = ((tc[0] * tc[0] + tc[1] * tc[1]) + tc[2] * tc[2]) + 1.0;
    if (nc[27] < ne[27]) {
      memcpy(q, qc, sizeof(qc)); memcpy(t, tc, sizeof(tc)); memcpy(ne, nc, sizeof(nc));
      memcpy(X, Xc, (size_t)m * 3 * sizeof(double));
      lambda = lambda_acc;
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      ba_normal_eq(&pb, q, t, lambda, ne, scratch);
    }
    if (dd <= 1.4e-14 * tt) break;
  }

  /* ---- outliers: a word any of whose edges ends with chi2 > delta^2 ----------------------------------------------- */
  double R2[9];
  sfo_quat_to_R(q, R2);
  int n_out = 0;
  const double lim = pb.delta * pb.delta;
  for (int i = 0; i < m; ++i) {
    int bad = 0;
    for (int cam = 0; cam < 2; ++cam) {
      const double* R = cam == 0 ? pb.R1 : R2;
      const double* tt = cam == 0 ? pb.t1 : t;
      const double* Xi = X + 3 * i;
      const double P[3] = {((R[0] * Xi[0] + R[1] * Xi[1]) + R[2] * Xi[2]) + tt[0],
                           ((R[3] * Xi[0] + R[4] * Xi[1]) + R[5] * Xi[2]) + tt[1],
                           ((R[6] * Xi[0] + R[7] * Xi[1]) + R[8] * Xi[2]) + tt[2]};
      double r[3], Ja[3], Jb[3], Js[3];
      const int rows = ba_edge(pb.fx, pb.fy, pb.b, P, (cam == 0 ? o1 : o2) + 3 * i, r, Ja, Jb, Js);
      if (rows == 0) { bad = 1; continue; }
      double chi2 = r[0] * r[0] + r[1] * r[1];
      if (rows == 3) chi2 = chi2 + r[2] * r[2];
      chi2 = chi2 * pb.info;
      if (chi2 > lim) bad = 1;
    }
    if (bad) { ++n_out; if (mask_out) mask_out[orig[i]] = 0; }
  }
  *n_inliers = m - n_out;
  if (*n_inliers < p->min_inliers) {
    *is_null = 1;      /* :1331-1336 */
    memset(T, 0, 12 * sizeof(float));
  } else {
    /* transforms[0] = optimizedPoses[2] = (L M)^-1 in float, as the PnP branch converts its pose */
    double Rd[9];
    sfo_quat_to_R(q, Rd);
    float Rf[9], tf[3], MR[9], Mt[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)Rd[i];
    for (int i = 0; i < 3; ++i) tf[i] = (float)t[i];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) MR[3 * i + j] = (L[4 * i] * Rf[j] + L[4 * i + 1] * Rf[3 + j]) + L[4 * i + 2] * Rf[6 + j];
      Mt[i] = ((L[4 * i] * tf[0] + L[4 * i + 1] * tf[1]) + L[4 * i + 2] * tf[2]) + L[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T[4 * i + j] = MR[3 * j + i];
      T[4 * i + 3] = -((MR[i] * Mt[0] + MR[3 + i] * Mt[1]) + MR[6 + i] * Mt[2]);
    }
    int allz = 1;
    for (int i = 0; i < 12; ++i) if (T[i] != 0.0f) allz = 0;
    *is_null = allz;
  }
  free(X); free(Xc); free(o1); free(o2); free(orig); free(scratch);
  return SF_OK;
}
