/*
 * sf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See sf_oracle.h for the
 * parity status ("NN pinned by scipy/numpy golden vectors; matching/RANSAC PARITY UNPINNED").
 *
 * Plain C99, single translation unit, no dependencies beyond libm.  Compile with
 * -ffp-contract=off: the floating-point operation ORDER written here is the canonical one the
 * HIP kernels restate independently (DESIGN.md "Canonical arithmetic"), so results can be
 * compared bit for bit.
 *
 * Citations: PKG = /root/reference/ros_ws/src/multi_robot_separators.
 * [upstream] marks semantics of un-vendored third-party code (rtabmap / PCL / OpenCV / FLANN)
 * restated from their published sources.
 */
#include "sf_oracle.h"
#include "sf_oracle_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


/* ========================================================================================= */
/* NN stage: DataHandler.find_matches, PKG/scripts/data_handler.py:166-209                    */
/* ========================================================================================= */

typedef struct { double v; int32_t i; } sfo_keyed;

static int sfo_keyed_cmp(const void* a, const void* b) {
  const sfo_keyed* x = (const sfo_keyed*)a;
  const sfo_keyed* y = (const sfo_keyed*)b;
  if (x->v < y->v) return -1;
  if (x->v > y->v) return 1;
  return (x->i > y->i) - (x->i < y->i); /* ties: lowest index first (numpy leaves it unspecified) */
}

int sfo_find_matches(const double* local, int n_l, const double* received, int n_r, int dim,
                     const int32_t* local_used, int n_local_used,
                     const int32_t* other_used, int n_other_used,
                     const int32_t* ignored_pairs, int n_ignored,
                     double netvlad_distance, int max_matches_nb,
                     sf_match* out, int cap, int* n_out,
                     double* row_min, int32_t* row_arg) {
  if (n_out) *n_out = 0;
  if (n_l <= 0 || n_r <= 0 || dim <= 0) return SF_EINVAL; /* guarded at data_handler.py:308 */
  double* dist = (double*)malloc((size_t)n_l * (size_t)n_r * sizeof(double));
  sfo_keyed* rows = (sfo_keyed*)malloc((size_t)n_l * sizeof(sfo_keyed));
  int32_t* arg = (int32_t*)malloc((size_t)n_l * sizeof(int32_t));
  if (!dist || !rows || !arg) { free(dist); free(rows); free(arg); return SF_ENOMEM; }

  /* :170 distances = cdist(local_descs, received_descs)  -- Euclidean, float64, direct form.
   * (rows are independent; OpenMP only changes who computes a row, never the arithmetic) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < n_l; ++i) {
    const double* a = local + (size_t)i * dim;
    for (int j = 0; j < n_r; ++j) {
      const double* b = received + (size_t)j * dim;
      double s = 0.0;
      for (int k = 0; k < dim; ++k) { double d = a[k] - b[k]; s += d * d; }
      dist[(size_t)i * n_r + j] = sqrt(s);
    }
  }
  /* :178-181 rows / columns of already-used keyframes -> inf */
  for (int u = 0; u < n_local_used; ++u) {
    int r = local_used[u];
    if (r < 0) r += n_l; /* numpy negative indexing */
    if (r < 0 || r >= n_l) { free(dist); free(rows); free(arg); return SF_ERANGE; }
    for (int j = 0; j < n_r; ++j) dist[(size_t)r * n_r + j] = INFINITY;
  }
  for (int u = 0; u < n_other_used; ++u) {
    int c = other_used[u];
    if (c < 0) c += n_r;
    if (c < 0 || c >= n_r) { free(dist); free(rows); free(arg); return SF_ERANGE; }
    for (int i = 0; i < n_l; ++i) dist[(size_t)i * n_r + c] = INFINITY;
  }
  /* :183-184 ignored pairs -> inf */
  for (int u = 0; u < n_ignored; ++u) {
    int r = ignored_pairs[2 * u], c = ignored_pairs[2 * u + 1];
    if (r < 0) r += n_l;
    if (c < 0) c += n_r;
    if (r < 0 || r >= n_l || c < 0 || c >= n_r) { free(dist); free(rows); free(arg); return SF_ERANGE; }
    dist[(size_t)r * n_r + c] = INFINITY;
  }
  /* :187-189 per-row arg-min (np.argsort(...)[:,0]) and its value */
  for (int i = 0; i < n_l; ++i) {
    int best = 0;
    double bv = dist[(size_t)i * n_r];
    for (int j = 1; j < n_r; ++j) {
      double v = dist[(size_t)i * n_r + j];
      if (v < bv) { bv = v; best = j; }
    }
    arg[i] = best;
    rows[i].v = bv;
    rows[i].i = i;
    if (row_min) row_min[i] = bv;
    if (row_arg) row_arg[i] = best;
  }
  /* :191 indexes_smallest_values_all_frames = argsort(smallest_values_each_frame) */
  qsort(rows, (size_t)n_l, sizeof(sfo_keyed), sfo_keyed_cmp);
  /* :193-205 walk */
  int n = 0;
  int lim = n_l < max_matches_nb ? n_l : max_matches_nb;
  for (int s = 0; s < lim; ++s) {
    int idx_local = rows[s].i;
    int idx_other = arg[idx_local];
    int taken = 0;
    for (int m = 0; m < n; ++m) if (out[m].idx_other == idx_other) { taken = 1; break; }
    if (taken) continue;                                       /* :199-200 */
    double d = dist[(size_t)idx_local * n_r + idx_other];
    if (d < netvlad_distance) {                                /* :202 */
      if (n < cap) { out[n].idx_local = idx_local; out[n].idx_other = idx_other; out[n].distance = d; }
      ++n;
      if (n > cap) { n = cap; break; }
    } else {
      break;                                                   /* :204-205 */
    }
  }
  if (n_out) *n_out = n;
  free(dist); free(rows); free(arg);
  return SF_OK;
}

/* ========================================================================================= */
/* Canonical arithmetic helpers                                                               */
/* ========================================================================================= */

/* Deterministic natural log from IEEE +,-,*,/ only (the HIP side restates the same series,
 * so the adaptive-stop bound k compares identically).  |error| ~ 1e-16. */
double sfo_canon_log(double x) {
  int e;
  double m = frexp(x, &e); /* x = m * 2^e, m in [0.5,1) */
  if (m < 0.70710678118654752440) { m = m * 2.0; e -= 1; }
  double z = (m - 1.0) / (m + 1.0);
  double z2 = z * z;
  /* sum_{k=0..13} z2^k / (2k+1), Horner from the top */
  double s = 1.0 / 27.0;
  for (int k = 12; k >= 0; --k) s = s * z2 + 1.0 / (double)(2 * k + 1);
  return 2.0 * z * s + (double)e * 0.69314718055994530942;
}

uint64_t sfo_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Stateless stand-in for PCL's drawIndexSample [upstream pcl/sample_consensus/sac_model.h]:
 * three distinct indices in [0,m), uniform, keyed by (seed, iteration, attempt). */
void sfo_sample_triplet(uint64_t seed, uint32_t iteration, uint32_t attempt, uint32_t m,
                        uint32_t out[3]) {
  uint64_t ha = sfo_mix(seed ^ sfo_mix(((uint64_t)iteration << 32) | (uint64_t)attempt));
  uint64_t hb = sfo_mix(ha);
  uint32_t r0 = (uint32_t)(ha >> 32), r1 = (uint32_t)ha, r2 = (uint32_t)(hb >> 32);
  uint32_t i0 = (uint32_t)(((uint64_t)r0 * (uint64_t)m) >> 32);
  uint32_t i1 = (uint32_t)(((uint64_t)r1 * (uint64_t)(m - 1)) >> 32);
  if (i1 >= i0) ++i1;
  uint32_t i2 = (uint32_t)(((uint64_t)r2 * (uint64_t)(m - 2)) >> 32);
  uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
  if (i2 >= lo) ++i2;
  if (i2 >= hi) ++i2;
  out[0] = i0; out[1] = i1; out[2] = i2;
}


static double sfo_det3(double a, double b, double c, double d, double e, double f, double g, double h,
                       double i) {
  return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}

/* Eigenvalues (descending, clamped at 0) of a symmetric positive semi-definite 3x3 matrix:
 * Newton on the characteristic cubic from the upper bound trace(C) (monotone convergence to the
 * largest root), then the deflated quadratic.  Only + - * / sqrt. */
static void sfo_sym3_eigenvalues(double C[4][4], double ev[3]) {
  const double c2 = (C[0][0] + C[1][1]) + C[2][2];
  const double c1 = ((C[0][0] * C[1][1] - C[0][1] * C[0][1]) + (C[0][0] * C[2][2] - C[0][2] * C[0][2])) +
                    (C[1][1] * C[2][2] - C[1][2] * C[1][2]);
  const double c0 = sfo_det3(C[0][0], C[0][1], C[0][2], C[0][1], C[1][1], C[1][2], C[0][2], C[1][2], C[2][2]);
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

/* Rotation + translation from the 3x3 cross-covariance S[j][k] = sum a_j b_k (a = demeaned
 * source, b = demeaned target), the two means and the spreads ga = sum |a|^2, gb = sum |b|^2:
 * Horn's unit-quaternion solution of the absolute-orientation problem.  Same optimum as the SVD
 * form PCL uses [upstream pcl::SampleConsensusModelRegistration::estimateRigidTransformationSVD ->
 * pcl::umeyama(src, tgt, false), computed in double].
 * The dominant eigenpair of Horn's 4x4 matrix N is found without an eigen-decomposition:
 * Newton's iteration on the characteristic quartic from the upper bound (ga+gb)/2 (monotone
 * convergence to the largest root), then the eigenvector as the best-conditioned column of
 * adj(N - lambda I).  Only + - * / sqrt: the HIP kernel restates it bit for bit.  A vanishing
 * adjugate (repeated top eigenvalue / no spread) falls back to the cyclic Jacobi solver. */
static void sfo_rigid_from_moments(const double S[3][3], const double mp[3], const double mq[3],
                                   double ga, double gb, double R[9], double t[3]) {
  double N[4][4], V[4][4];
  N[0][0] = (S[0][0] + S[1][1]) + S[2][2];
  N[1][1] = (S[0][0] - S[1][1]) - S[2][2];
  N[2][2] = (S[1][1] - S[0][0]) - S[2][2];
  N[3][3] = (S[2][2] - S[0][0]) - S[1][1];
  N[0][1] = N[1][0] = S[1][2] - S[2][1];
  N[0][2] = N[2][0] = S[2][0] - S[0][2];
  N[0][3] = N[3][0] = S[0][1] - S[1][0];
  N[1][2] = N[2][1] = S[0][1] + S[1][0];
  N[1][3] = N[3][1] = S[2][0] + S[0][2];
  N[2][3] = N[3][2] = S[1][2] + S[2][1];

  /* p(x) = x^4 + c2 x^2 + c1 x + c0  (N is traceless) */
  double ss = 0.0;
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) ss += S[j][k] * S[j][k];
  const double c2 = -2.0 * ss;
  const double c1 = -8.0 * sfo_det3(S[0][0], S[0][1], S[0][2], S[1][0], S[1][1], S[1][2], S[2][0], S[2][1], S[2][2]);
  const double m0 = sfo_det3(N[1][1], N[1][2], N[1][3], N[2][1], N[2][2], N[2][3], N[3][1], N[3][2], N[3][3]);
  const double m1 = sfo_det3(N[1][0], N[1][2], N[1][3], N[2][0], N[2][2], N[2][3], N[3][0], N[3][2], N[3][3]);
  const double m2 = sfo_det3(N[1][0], N[1][1], N[1][3], N[2][0], N[2][1], N[2][3], N[3][0], N[3][1], N[3][3]);
  const double m3 = sfo_det3(N[1][0], N[1][1], N[1][2], N[2][0], N[2][1], N[2][2], N[3][0], N[3][1], N[3][2]);
  const double c0 = ((N[0][0] * m0 - N[0][1] * m1) + N[0][2] * m2) - N[0][3] * m3;

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

  /* B = N - x I ; adj(B) is symmetric: 10 cofactors */
  const double b00 = N[0][0] - x, b11 = N[1][1] - x, b22 = N[2][2] - x, b33 = N[3][3] - x;
  const double b01 = N[0][1], b02 = N[0][2], b03 = N[0][3], b12 = N[1][2], b13 = N[1][3], b23 = N[2][3];
  const double a00 = sfo_det3(b11, b12, b13, b12, b22, b23, b13, b23, b33);
  const double a11 = sfo_det3(b00, b02, b03, b02, b22, b23, b03, b23, b33);
  const double a22 = sfo_det3(b00, b01, b03, b01, b11, b13, b03, b13, b33);
  const double a33 = sfo_det3(b00, b01, b02, b01, b11, b12, b02, b12, b22);
  const double a01 = -sfo_det3(b01, b12, b13, b02, b22, b23, b03, b23, b33);
  const double a02 = sfo_det3(b01, b11, b13, b02, b12, b23, b03, b13, b33);
  const double a03 = -sfo_det3(b01, b11, b12, b02, b12, b22, b03, b13, b23);
  const double a12 = -sfo_det3(b00, b01, b03, b02, b12, b23, b03, b13, b33);
  const double a13 = sfo_det3(b00, b01, b02, b02, b12, b22, b03, b13, b23);
  const double a23 = -sfo_det3(b00, b01, b02, b01, b11, b12, b03, b13, b23);
  double best = fabs(a00);
  double w = a00, qx = a01, qy = a02, qz = a03;
  if (fabs(a11) > best) { best = fabs(a11); w = a01; qx = a11; qy = a12; qz = a13; }
  if (fabs(a22) > best) { best = fabs(a22); w = a02; qx = a12; qy = a22; qz = a23; }
  if (fabs(a33) > best) { best = fabs(a33); w = a03; qx = a13; qy = a23; qz = a33; }
  double nrm2 = ((w * w + qx * qx) + qy * qy) + qz * qz;
  if (!(best > 0.0) || !(nrm2 > 0.0) || !isfinite(nrm2)) {
    /* vanishing adjugate (an exactly degenerate configuration: the largest eigenvalue is repeated, or N = 0): a
     * vector of its eigenspace by 64 steps of the power iteration on N + shift I (positive semi-definite with
     * shift = (ga + gb) / 2 >= |eigenvalues|), from a fixed start.  Round 1 ran a cyclic Jacobi here; its two 4x4
     * matrices were what pushed the GPU kernels into scratch on the HOT path, although this branch is practically
     * never taken -- any unit vector of the eigenspace is a valid answer, and this one needs 14 doubles. */
    (void)V;
    const double shift = 0.5 * (ga + gb);
    double v0 = 1.0, v1 = 0.5, v2 = 0.25, v3 = 0.125;
    for (int it = 0; it < 64; ++it) {
      const double u0 = (((N[0][0] + shift) * v0 + N[0][1] * v1) + N[0][2] * v2) + N[0][3] * v3;
      const double u1 = ((N[0][1] * v0 + (N[1][1] + shift) * v1) + N[1][2] * v2) + N[1][3] * v3;
      const double u2 = ((N[0][2] * v0 + N[1][2] * v1) + (N[2][2] + shift) * v2) + N[2][3] * v3;
      const double u3 = ((N[0][3] * v0 + N[1][3] * v1) + N[2][3] * v2) + (N[3][3] + shift) * v3;
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
  R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
  R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
  R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
  for (int j = 0; j < 3; ++j)
    t[j] = mq[j] - ((R[3 * j] * mp[0] + R[3 * j + 1] * mp[1]) + R[3 * j + 2] * mp[2]);
}

/* Sequential-order rigid fit (used for the 3-point RANSAC hypotheses). src/dst: n x 3 double */
void sfo_fit_rigid(const double* src, const double* dst, int n, double R[9], double t[3]) {
  double inv_n = 1.0 / (double)n;
  double mp[3] = {0, 0, 0}, mq[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) { mp[j] += src[3 * i + j]; mq[j] += dst[3 * i + j]; }
  for (int j = 0; j < 3; ++j) { mp[j] *= inv_n; mq[j] *= inv_n; }
  double S[3][3] = {{0}};
  double ga = 0.0, gb = 0.0;
  for (int i = 0; i < n; ++i) {
    double a[3], b[3];
    for (int j = 0; j < 3; ++j) { a[j] = src[3 * i + j] - mp[j]; b[j] = dst[3 * i + j] - mq[j]; }
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) S[j][k] += a[j] * b[k];
    ga += (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    gb += (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
  }
  sfo_rigid_from_moments(S, mp, mq, ga, gb, R, t);
}

/* Canonical block sum: SFO_LANES strided partials, xor-butterfly inside each group of 64,
 * then the four group sums folded left to right (what a 256-thread workgroup computes). */
double sfo_block_sum(const double* x, int n) {
  double part[SFO_LANES], tmp[SFO_LANES];
  for (int l = 0; l < SFO_LANES; ++l) {
    double s = 0.0;
    for (int i = l; i < n; i += SFO_LANES) s += x[i];
    part[l] = s;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < SFO_LANES; ++l) tmp[l] = part[l] + part[l ^ off];
    memcpy(part, tmp, sizeof(part));
  }
  return ((part[0] + part[64]) + part[128]) + part[192];
}

/* Block-order rigid fit over the members of `mask` (used by the refine loop:
 * [upstream pcl::SampleConsensusModelRegistration::optimizeModelCoefficients]). */
static void sfo_fit_rigid_masked(const float* src, const float* dst, int m, const uint8_t* mask,
                                 int n_in, double R[9], double t[3], double* scratch) {
  double inv_n = 1.0 / (double)n_in;
  double mp[3], mq[3];
  for (int j = 0; j < 3; ++j) {
    for (int i = 0; i < m; ++i) scratch[i] = mask[i] ? (double)src[3 * i + j] : 0.0;
    mp[j] = sfo_block_sum(scratch, m) * inv_n;
    for (int i = 0; i < m; ++i) scratch[i] = mask[i] ? (double)dst[3 * i + j] : 0.0;
    mq[j] = sfo_block_sum(scratch, m) * inv_n;
  }
  double S[3][3];
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 3; ++k) {
      for (int i = 0; i < m; ++i)
        scratch[i] = mask[i] ? ((double)src[3 * i + j] - mp[j]) * ((double)dst[3 * i + k] - mq[k]) : 0.0;
      S[j][k] = sfo_block_sum(scratch, m);
    }
  double g[2];
  for (int w = 0; w < 2; ++w) {
    const float* pts = w == 0 ? src : dst;
    const double* mean = w == 0 ? mp : mq;
    for (int i = 0; i < m; ++i) {
      if (mask[i]) {
        double d0 = (double)pts[3 * i] - mean[0], d1 = (double)pts[3 * i + 1] - mean[1],
               d2 = (double)pts[3 * i + 2] - mean[2];
        scratch[i] = (d0 * d0 + d1 * d1) + d2 * d2;
      } else {
        scratch[i] = 0.0;
      }
    }
    g[w] = sfo_block_sum(scratch, m);
  }
  sfo_rigid_from_moments(S, mp, mq, g[0], g[1], R, t);
}

/* squared residual of one correspondence under float coefficients c[12] (row-major 3x4):
 * [upstream pcl::SampleConsensusModelRegistration::selectWithinDistance: float Matrix4f * Vector4f,
 *  (p_tr - pt_tgt).squaredNorm()] with the canonical fma chain. */
float sfo_residual2(const float c[12], const float* p, const float* q) {
  float px = fmaf(c[2], p[2], fmaf(c[1], p[1], fmaf(c[0], p[0], c[3])));
  float py = fmaf(c[6], p[2], fmaf(c[5], p[1], fmaf(c[4], p[0], c[7])));
  float pz = fmaf(c[10], p[2], fmaf(c[9], p[1], fmaf(c[8], p[0], c[11])));
  float dx = px - q[0], dy = py - q[1], dz = pz - q[2];
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

int sfo_finite3(const float* p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }
static int sfo_nonzero3(const float* p) { return p[0] != 0.0f || p[1] != 0.0f || p[2] != 0.0f; }

static int sfo_popcount_row(const uint8_t* a, const uint8_t* b, int cols) {
  int d = 0, c = 0;
  for (; c + 8 <= cols; c += 8) {
    uint64_t x, y;
    memcpy(&x, a + c, 8); memcpy(&y, b + c, 8);
    d += __builtin_popcountll(x ^ y);
  }
  for (; c < cols; ++c) d += __builtin_popcount((unsigned)(a[c] ^ b[c]));
  return d;
}

/* ========================================================================================= */
/* Global matching: PKG/src/myRegistrationVis.cpp:826-895                                      */
/* ========================================================================================= */
/* desc_type 1 (float32 rows, `cols` = 4 * dimensions): squared L2 distance accumulated in float32 in dimension order,
 * multiply and add unfused (-ffp-contract=off) -- the canonical order the device kernels follow (csrc/k_match.hip
 * knn2_scan_l2).  [upstream] VWDictionary::addNewWords compares squared L2 distances of float descriptors;
 * cv::BFMatcher(NORM_L2) in the guided branch reports their square root. */
static float sfo_l2sq_row(const uint8_t* a, const uint8_t* b, int cols) {
  const float* x = (const float*)a;
  const float* y = (const float*)b;
  float s = 0.0f;
  for (int k = 0; k < cols / 4; ++k) { const float d = x[k] - y[k]; s = s + d * d; }
  return s;
}

int sfo_match_global(const uint8_t* desc_from, int k_from, const uint8_t* desc_to, int k_to,
                     int cols, float nndr, int has3d_from, int has3d_to,
                     uint16_t* corr_from, uint16_t* corr_to, int* n_corr,
                     int* n_words_from, int* n_words_to, int* n_words_to_2d, int desc_type) {
  *n_corr = 0; *n_words_from = 0; *n_words_to = 0; *n_words_to_2d = 0;
  if (k_from <= 0 || k_to <= 0) {
    /* :897-910 only fake "from" words; wordsTo stays empty */
    *n_words_from = (k_from > 0 && has3d_from) ? k_from : 0;
    return SF_OK;
  }
  int32_t* match = (int32_t*)malloc((size_t)k_to * sizeof(int32_t));
  int32_t* cnt = (int32_t*)calloc((size_t)k_from, sizeof(int32_t));
  int32_t* owner = (int32_t*)malloc((size_t)k_from * sizeof(int32_t));
  if (!match || !cnt || !owner) { free(match); free(cnt); free(owner); return SF_ENOMEM; }
  /* :839-852 dictionary of the "from" words (ids 0..k_from-1), addNewWords(descriptorsTo):
   * [upstream VWDictionary::addNewWords, brute force] kNN k=2 (cv::BFMatcher NORM_HAMMING),
   * accept nearest id unless d1 > nndr * d2 or fewer than 2 results, else mint a new id. */
  for (int t = 0; t < k_to; ++t) {
    const uint8_t* q = desc_to + (size_t)t * cols;
    int d1 = 1 << 30, d2 = 1 << 30, i1 = -1;
    for (int f = 0; f < k_from && desc_type != 1; ++f) {
      int d = sfo_popcount_row(q, desc_from + (size_t)f * cols, cols);
      if (d < d1) { d2 = d1; d1 = d; i1 = f; }
      else if (d < d2) { d2 = d; }
    }
    int accept = 0;
    if (desc_type == 1) {
      /* float32 rows: the same scan on squared L2 distances (strict comparisons: ties keep the lower id) */
      float f1 = INFINITY, f2 = INFINITY;
      i1 = -1;
      for (int f = 0; f < k_from; ++f) {
        const float d = sfo_l2sq_row(q, desc_from + (size_t)f * cols, cols);
        if (d < f1) { f2 = f1; f1 = d; i1 = f; }
        else if (d < f2) { f2 = d; }
      }
      if (k_from >= 2 && i1 >= 0) accept = !(f1 > nndr * f2);
    } else
    if (k_from >= 2) accept = !((float)d1 > nndr * (float)d2);
    match[t] = accept ? i1 : -1;
    if (accept) { cnt[i1]++; owner[i1] = t; }
  }
  /* :856-894 keep ids occurring exactly once on each side.  "from" ids are distinct by
   * construction; a "from" id matched by two "to" rows is dropped on the "to" side. */
  int n = 0, unique_to = 0;
  for (int t = 0; t < k_to; ++t) if (match[t] < 0 || cnt[match[t]] == 1) ++unique_to;
  for (int f = 0; f < k_from; ++f)
    if (cnt[f] == 1) { corr_from[n] = (uint16_t)f; corr_to[n] = (uint16_t)owner[f]; ++n; }
  *n_corr = n;
  *n_words_from = has3d_from ? k_from : 0;   /* :870-873 */
  *n_words_to_2d = unique_to;                /* :886 */
  *n_words_to = has3d_to ? unique_to : 0;    /* :888-891 */
  free(match); free(cnt); free(owner);
  return SF_OK;
}

/* ========================================================================================= */
/* Guided matching: PKG/src/myRegistrationVis.cpp:476-825 (sub-branch :667-818)                */
/* ========================================================================================= */
static int sfo_octave(int32_t o) { int v = o & 255; return v < 128 ? v : (-128 | v); } /* :709-710 */

int sfo_match_guided(const sf_params* p, const float* guess,
                     const uint8_t* desc_from, const float* xyz_from, const sf_keypoint* kp_from,
                     int k_from,
                     const uint8_t* desc_to, const sf_keypoint* kp_to, int k_to, int has3d_to,
                     int cols,
                     uint16_t* corr_from, uint16_t* corr_to, int* n_corr,
                     int* n_words_from, int* n_words_to, int* n_words_to_2d, int* all_outside) {
  *n_corr = 0; *n_words_from = 0; *n_words_to = 0; *n_words_to_2d = 0; *all_outside = 0;
  const float* L = p->local_transform;
  /* :486-487 guessCameraRef = (guess * localTransform).inverse()   (float) */
  float GR[9], Gt[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      GR[3 * i + j] = (guess[4 * i] * L[j] + guess[4 * i + 1] * L[4 + j]) + guess[4 * i + 2] * L[8 + j];
    Gt[i] = ((guess[4 * i] * L[3] + guess[4 * i + 1] * L[7]) + guess[4 * i + 2] * L[11]) + guess[4 * i + 3];
  }
  float Rc[9], tc[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rc[3 * i + j] = GR[3 * j + i];
  for (int i = 0; i < 3; ++i)
    tc[i] = -((Rc[3 * i] * Gt[0] + Rc[3 * i + 1] * Gt[1]) + Rc[3 * i + 2] * Gt[2]);
  double Rd[9], td[3];
  for (int i = 0; i < 9; ++i) Rd[i] = (double)Rc[i];
  for (int i = 0; i < 3; ++i) td[i] = (double)tc[i];

  float* pu = (float*)malloc((size_t)(k_from > 0 ? k_from : 1) * sizeof(float));
  float* pv = (float*)malloc((size_t)(k_from > 0 ? k_from : 1) * sizeof(float));
  uint8_t* inimg = (uint8_t*)calloc((size_t)(k_from > 0 ? k_from : 1), 1);
  int32_t* claim = (int32_t*)malloc((size_t)(k_to > 0 ? k_to : 1) * sizeof(int32_t));
  if (!pu || !pv || !inimg || !claim) { free(pu); free(pv); free(inimg); free(claim); return SF_ENOMEM; }
  for (int t = 0; t < k_to; ++t) claim[t] = -1;

  /* :488-514 project kptsFrom3D (cv::projectPoints, double, no distortion), keep points inside
   * the image with positive depth */
  const float wlim = (float)(p->image_width - 1), hlim = (float)(p->image_height - 1);
  int n_proj = 0, n_finite = 0;
  for (int i = 0; i < k_from; ++i) {
    const float* P = xyz_from + 3 * (size_t)i;
    if (!sfo_finite3(P)) continue;
    ++n_finite;
    float zf = ((Rc[6] * P[0] + Rc[7] * P[1]) + Rc[8] * P[2]) + tc[2]; /* util3d::transformPoint, :507 */
    double X = ((Rd[0] * (double)P[0] + Rd[1] * (double)P[1]) + Rd[2] * (double)P[2]) + td[0];
    double Y = ((Rd[3] * (double)P[0] + Rd[4] * (double)P[1]) + Rd[5] * (double)P[2]) + td[1];
    double Z = ((Rd[6] * (double)P[0] + Rd[7] * (double)P[1]) + Rd[8] * (double)P[2]) + td[2];
    double iz = (Z != 0.0) ? 1.0 / Z : 1.0;
    float u = (float)((X * iz) * p->fx + p->cx);
    float v = (float)((Y * iz) * p->fy + p->cy);
    int ok = isfinite(u) && isfinite(v) && !(u < 0.0f) && !(u >= wlim) && !(v < 0.0f) && !(v >= hlim) &&
             (zf > 0.0f);
    if (ok) { pu[i] = u; pv[i] = v; inimg[i] = 1; ++n_proj; }
  }
  *n_words_from = n_finite; /* :766-774 + :793-802: every finite "from" point becomes a word */
  if (n_proj == 0) {        /* :820-823 */
    *all_outside = 1;
    *n_words_from = 0;
    free(pu); free(pv); free(inimg); free(claim);
    return SF_OK;
  }
  /* :670-789 radius search around each projection, octave filter, BF Hamming k=2 + NNDR,
   * single candidate accepted without descriptor test, each "to" index claimed once */
  const float r2 = (float)p->guess_win_size * (float)p->guess_win_size;
  for (int i = 0; i < k_from; ++i) {
    if (!inimg[i]) continue;
    int octf = sfo_octave(kp_from[i].octave);
    const uint8_t* q = desc_from + (size_t)i * cols;
    int oi = 0, d0 = 1 << 30, d1 = 1 << 30, i0 = -1, last = -1;
    float f0 = INFINITY, f1 = INFINITY;     /* desc_type 1: cv::BFMatcher(NORM_L2) distances */
    for (int t = 0; t < k_to; ++t) {
      float dx = pu[i] - kp_to[t].x, dy = pv[i] - kp_to[t].y;
      float d2 = dx * dx + dy * dy;
      if (!(d2 < r2)) continue;
      if (sfo_octave(kp_to[t].octave) != octf) continue;
      ++oi; last = t;
      if (p->desc_type == 1) {
        const float d = sqrtf(sfo_l2sq_row(q, desc_to + (size_t)t * cols, cols));
        if (d < f0) { f1 = f0; f0 = d; i0 = t; }
        else if (d < f1) { f1 = d; }
        continue;
      }
      int d = sfo_popcount_row(q, desc_to + (size_t)t * cols, cols);
      if (d < d0) { d1 = d0; d0 = d; i0 = t; }
      else if (d < d1) { d1 = d; }
    }
    int matched = -1;
    if (oi >= 2 && p->desc_type == 1) { if (i0 >= 0 && f0 < p->nndr * f1) matched = i0; }
    else
    if (oi >= 2) { if ((float)d0 < p->nndr * (float)d1) matched = i0; }   /* :744 */
    else if (oi == 1) matched = last;                                     /* :751-754, :756-764 */
    if (matched >= 0 && claim[matched] < 0) claim[matched] = i;           /* :776-787 */
  }
  int n = 0;
  /* ascending "from" id order: collect (claimer, t) pairs then sort by claimer */
  for (int i = 0; i < k_from; ++i) {
    if (!inimg[i]) continue;
    for (int t = 0; t < k_to; ++t)
      if (claim[t] == i) { corr_from[n] = (uint16_t)i; corr_to[n] = (uint16_t)t; ++n; break; }
  }
  *n_corr = n;
  *n_words_to_2d = k_to;                 /* :776-787 + :804-817 every "to" row gets an id */
  *n_words_to = has3d_to ? k_to : 0;
  free(pu); free(pv); free(inimg); free(claim);
  return SF_OK;
}

/* ========================================================================================= */
/* Motion estimation: util3d::estimateMotion3DTo3D [upstream], called at                       */
/* PKG/src/myRegistrationVis.cpp:1122-1131                                                     */
/* ========================================================================================= */

/* 2.1981 * median of the squared inlier residuals
 * [upstream pcl::SampleConsensusModel::computeVariance] */
static double sfo_variance(const double* d2, int n, double* scratch) {
  if (n <= 0) return NAN;
  memcpy(scratch, d2, (size_t)n * sizeof(double));
  int med = n >> 1;
  /* std::nth_element: quickselect of the med-th smallest (only its VALUE matters) */
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    double pivot = scratch[lo + ((hi - lo) >> 1)];
    int i = lo, j = hi;
    while (i <= j) {
      while (scratch[i] < pivot) ++i;
      while (scratch[j] > pivot) --j;
      if (i <= j) { double tmp = scratch[i]; scratch[i] = scratch[j]; scratch[j] = tmp; ++i; --j; }
    }
    if (med <= j) hi = j;
    else if (med >= i) lo = i;
    else break;
  }
  return 2.1981 * scratch[med];
}

/* selectWithinDistance: mask + squared distances of the selected set (ascending index) */
static int sfo_select(const float c[12], const float* src, const float* dst, int m, double thr2,
                      uint8_t* mask, double* d2_list) {
  int n = 0;
  for (int i = 0; i < m; ++i) {
    float r2 = sfo_residual2(c, src + 3 * i, dst + 3 * i);
    int in = (double)r2 < thr2;
    mask[i] = (uint8_t)in;
    if (in) d2_list[n++] = (double)r2;
  }
  return n;
}

int sfo_estimate_motion_3d3d(const sf_params* p,
                             const float* xyz_from, const float* xyz_to,
                             const uint16_t* corr_from, const uint16_t* corr_to, int n_corr,
                             sfo_motion* out, uint8_t* inlier_mask_out) {
  memset(out, 0, sizeof(*out));
  out->is_null = 1;
  out->variance = 1.0;            /* *covariance = eye(6) */
  out->variance_ang = 1.0;
  out->ransac_best_iteration = -1;

  /* [upstream util3d::findCorrespondences] id-matched, finite and non-zero 3D pairs,
   * ascending id.  src = "from" (PCL model input_), dst = "to" (PCL target_). */
  int cap = n_corr > 0 ? n_corr : 1;
  float* src = (float*)malloc((size_t)cap * 3 * sizeof(float));
  float* dst = (float*)malloc((size_t)cap * 3 * sizeof(float));
  int32_t* orig = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
  uint8_t* mask_a = (uint8_t*)calloc((size_t)cap, 1);
  uint8_t* mask_b = (uint8_t*)calloc((size_t)cap, 1);
  double* d2_list = (double*)malloc((size_t)cap * sizeof(double));
  double* scratch = (double*)malloc((size_t)cap * sizeof(double));
  int32_t* counts = (int32_t*)malloc((size_t)(p->iterations + 2) * sizeof(int32_t));
  if (!src || !dst || !orig || !mask_a || !mask_b || !d2_list || !scratch || !counts) {
    free(src); free(dst); free(orig); free(mask_a); free(mask_b); free(d2_list); free(scratch); free(counts);
    return SF_ENOMEM;
  }
  int m = 0;
  for (int i = 0; i < n_corr; ++i) {
    const float* a = xyz_from + 3 * (size_t)corr_from[i];
    const float* b = xyz_to + 3 * (size_t)corr_to[i];
    if (sfo_finite3(a) && sfo_finite3(b) && sfo_nonzero3(a) && sfo_nonzero3(b)) {
      memcpy(src + 3 * m, a, 12); memcpy(dst + 3 * m, b, 12); orig[m] = i; ++m;
    }
  }
  out->matches = m;
  if (inlier_mask_out) memset(inlier_mask_out, 0, (size_t)n_corr);

  float coef[12];
  int n_inliers = 0;
  int have_model = 0;

  if (m >= p->min_inliers && m >= 3) {
    /* ---- [upstream pcl::SampleConsensusModel::computeSampleDistanceThreshold] (double here) */
    double inv_m = 1.0 / (double)m;
    double mean[3];
    for (int j = 0; j < 3; ++j) {
      for (int i = 0; i < m; ++i) scratch[i] = (double)src[3 * i + j];
      mean[j] = sfo_block_sum(scratch, m) * inv_m;
    }
    double C[4][4];
    for (int j = 0; j < 3; ++j)
      for (int k = j; k < 3; ++k) {
        for (int i = 0; i < m; ++i)
          scratch[i] = ((double)src[3 * i + j] - mean[j]) * ((double)src[3 * i + k] - mean[k]);
        C[j][k] = C[k][j] = sfo_block_sum(scratch, m) * inv_m;
      }
    double ev[3];
    sfo_sym3_eigenvalues(C, ev);
    double sdt = ((sqrt(ev[0]) + sqrt(ev[1])) + sqrt(ev[2])) / 3.0;
    sdt = sdt * sdt;

    /* ---- [upstream pcl::RandomSampleConsensus::computeModel] -------------------------------- */
    const double thr = (double)p->inlier_distance;
    const double thr2 = thr * thr;
    const int max_it = p->iterations;
    /* evaluate hypotheses it = 0..max_it (PCL runs until iterations_ > max_iterations_) */
    int first_invalid = max_it + 1;
    for (int it = 0; it <= max_it; ++it) {
      uint32_t s[3];
      int good = 0;
      for (int a = 0; a < p->max_sample_checks; ++a) {
        sfo_sample_triplet(p->seed, (uint32_t)it, (uint32_t)a, (uint32_t)m, s);
        /* [upstream SampleConsensusModelRegistration::isSampleGood] on the source cloud */
        const float *p0 = src + 3 * s[0], *p1 = src + 3 * s[1], *p2 = src + 3 * s[2];
        float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        float cx = p2[0] - p1[0], cy = p2[1] - p1[1], cz = p2[2] - p1[2];
        float da = (ax * ax + ay * ay) + az * az;
        float db = (bx * bx + by * by) + bz * bz;
        float dc = (cx * cx + cy * cy) + cz * cz;
        if ((double)da > sdt && (double)db > sdt && (double)dc > sdt) { good = 1; break; }
      }
      if (!good) { counts[it] = -1; if (it < first_invalid) first_invalid = it; continue; }
      double ps[9], qs[9], R[9], t[3];
      for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j) { ps[3 * k + j] = (double)src[3 * s[k] + j]; qs[3 * k + j] = (double)dst[3 * s[k] + j]; }
      sfo_fit_rigid(ps, qs, 3, R, t);
      float c[12];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c[4 * i + j] = (float)R[3 * i + j]; c[4 * i + 3] = (float)t[i]; }
      int cnt = 0;
      for (int i = 0; i < m; ++i) cnt += ((double)sfo_residual2(c, src + 3 * i, dst + 3 * i) < thr2);
      counts[it] = cnt;
    }
    /* sequential scan reproducing PCL's loop (adaptive k) */
    double k = 1.0;
    const double log_probability = sfo_canon_log(1.0 - 0.99);
    const double one_over_m = 1.0 / (double)m;
    int best = -1, best_it = -1, it = 0;
    while (p->ransac_adaptive_stop ? ((double)it < k) : 1) {
      if (it > max_it) break;
      if (counts[it] < 0) break; /* getSamples failed -> PCL breaks out */
      if (counts[it] > best) {
        best = counts[it]; best_it = it;
        double w = (double)best * one_over_m;
        double pno = 1.0 - (w * w) * w;
        if (pno < DBL_EPSILON) pno = DBL_EPSILON;
        if (pno > 1.0 - DBL_EPSILON) pno = 1.0 - DBL_EPSILON;
        k = log_probability / sfo_canon_log(pno);
      }
      ++it;
      if (it > max_it) break;
    }
    out->ransac_iterations_run = it;
    out->ransac_best_iteration = best_it;
    out->ransac_best_count = best;

    if (best_it >= 0) {
      /* recompute the winning model */
      uint32_t s[3] = {0, 0, 0};
      for (int a = 0; a < p->max_sample_checks; ++a) {
        sfo_sample_triplet(p->seed, (uint32_t)best_it, (uint32_t)a, (uint32_t)m, s);
        const float *p0 = src + 3 * s[0], *p1 = src + 3 * s[1], *p2 = src + 3 * s[2];
        float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        float cx = p2[0] - p1[0], cy = p2[1] - p1[1], cz = p2[2] - p1[2];
        float da = (ax * ax + ay * ay) + az * az;
        float db = (bx * bx + by * by) + bz * bz;
        float dc = (cx * cx + cy * cy) + cz * cz;
        if ((double)da > sdt && (double)db > sdt && (double)dc > sdt) break;
      }
      double ps[9], qs[9], R[9], t[3];
      for (int kk = 0; kk < 3; ++kk)
        for (int j = 0; j < 3; ++j) { ps[3 * kk + j] = (double)src[3 * s[kk] + j]; qs[3 * kk + j] = (double)dst[3 * s[kk] + j]; }
      sfo_fit_rigid(ps, qs, 3, R, t);
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) coef[4 * i + j] = (float)R[3 * i + j]; coef[4 * i + 3] = (float)t[i]; }

      /* sac.getInliers(): selectWithinDistance(best, threshold) */
      uint8_t* inl = mask_a;      /* "inliers" */
      int n_inl = sfo_select(coef, src, dst, m, thr2, inl, d2_list);
      int n_last = n_inl;          /* size of error_sqr_dists_ */

      /* ---- refine loop [upstream rtabmap util3d::transformFromXYZCorrespondences, a copy of
       * pcl::SampleConsensus::refineModel] ------------------------------------------------ */
      if (p->refine_iterations > 0) {
        double error_threshold = thr;
        int refine_iterations = 0;
        int inlier_changed = 0, oscillating = 0;
        uint8_t* prev = mask_a;   /* prev_inliers = inliers */
        uint8_t* neu = mask_b;    /* new_inliers (empty)    */
        int n_prev = n_inl, n_new = 0;
        memset(neu, 0, (size_t)m);
        int sizes[64]; int n_sizes = 0;
        float newc[12]; memcpy(newc, coef, sizeof(newc));
        do {
          /* optimizeModelCoefficients(prev_inliers, new, new) */
          if (n_prev >= 3) {
            double R2[9], t2[3];
            sfo_fit_rigid_masked(src, dst, m, prev, n_prev, R2, t2, scratch);
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) newc[4 * i + j] = (float)R2[3 * i + j]; newc[4 * i + 3] = (float)t2[i]; }
          }
          if (n_sizes < 64) sizes[n_sizes] = n_prev;
          ++n_sizes;
          n_new = sfo_select(newc, src, dst, m, error_threshold * error_threshold, neu, d2_list);
          n_last = n_new;
          ++out->refine_rounds;
          if (n_new == 0) {
            ++refine_iterations;
            if (refine_iterations >= p->refine_iterations) break;
            continue;
          }
          double variance = sfo_variance(d2_list, n_new, scratch);
          double sthr = p->refine_sigma * sqrt(variance);
          error_threshold = thr < sthr ? thr : sthr;
          inlier_changed = 0;
          { uint8_t* tmp = prev; prev = neu; neu = tmp; int tn = n_prev; n_prev = n_new; n_new = tn; }
          if (n_new != n_prev) {
            if (n_sizes >= 4 && n_sizes <= 64) {
              if (sizes[n_sizes - 1] == sizes[n_sizes - 3] && sizes[n_sizes - 2] == sizes[n_sizes - 4]) {
                oscillating = 1;
                break;
              }
            }
            inlier_changed = 1;
            continue;
          }
          for (int i = 0; i < m; ++i) if (prev[i] != neu[i]) { inlier_changed = 1; break; }
        } while (inlier_changed && ++refine_iterations < p->refine_iterations);
        (void)oscillating;
        /* std::swap(inliers, new_inliers); model_coefficients = new_model_coefficients */
        inl = neu; n_inl = n_new;
        memcpy(coef, newc, sizeof(coef));
      }

      if (n_inl >= 3) {
        double variance = sfo_variance(d2_list, n_last, scratch); /* model->computeVariance() */
        out->variance = variance;                                 /* *covariance *= variance   */
        out->variance_ang = variance;
        have_model = 1;
        n_inliers = n_inl;
        if (inlier_mask_out) for (int i = 0; i < m; ++i) if (inl[i]) inlier_mask_out[orig[i]] = 1;
      }
    }
  }

  if (have_model) {
    out->inliers = n_inliers;
    if (n_inliers >= p->min_inliers) {
      /* transform.inverse(): pose of "to" in "from" (p_from = T p_to) */
      double R[9], t[3];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)coef[4 * i + j]; t[i] = (double)coef[4 * i + 3]; }
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out->transform[4 * i + j] = (float)R[3 * j + i];
        out->transform[4 * i + 3] = (float)(-((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]));
      }
      out->is_null = 0;
      /* rtabmap::Transform::isNull(): an all-zero matrix reads as null */
      int allz = 1;
      for (int i = 0; i < 12; ++i) if (out->transform[i] != 0.0f) allz = 0;
      if (allz) out->is_null = 1;
    }
  }
  free(src); free(dst); free(orig); free(mask_a); free(mask_b); free(d2_list); free(scratch); free(counts);
  return SF_OK;
}

/* ========================================================================================= */
/* Two-pass driver: PKG/src/stereoCamGeometricTools.cpp:122-178                                */
/* ========================================================================================= */

typedef struct {
  float transform[12];
  int is_null;
  double cov_diag;   /* covariance diagonal before the clamp: linear block ...   */
  double cov_diag_ang;   /* ... and angular block (equal for the 3D-3D estimator)  */
  int inliers, matches;
} sfo_pass;

static int sfo_is_identity(const float* T) {
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int i = 0; i < 12; ++i) if (T[i] != I[i]) return 0;
  return 1;
}

static int sfo_validate(const sf_features* f) {
  if (!f) return SF_EINVAL;
  if (f->rows > 0 && (!f->desc || f->cols == 0)) return SF_EINVAL;
  if (f->rows > SF_MAX_FEATURES) return SF_ERANGE;
  if (f->n3d != 0 && f->n3d != (int32_t)f->rows) return SF_EINVAL;  /* myRegistrationVis.cpp:859,878 */
  if (f->nkp != (int32_t)f->rows) return SF_EINVAL;                   /* :482,:879 */
  if (f->n3d > 0 && !f->xyz) return SF_EINVAL;
  if (f->nkp > 0 && !f->kpts) return SF_EINVAL;
  return SF_OK;
}

/* Transform::to3DoF [upstream rtabmap Transform.cpp]: Transform(x, y, 0, 0, 0, yaw) with (roll, pitch, yaw) from
 * pcl::getEulerAngles, i.e. yaw = atan2(r21, r11).  Canonical form shared with the device (csrc/sf_device_math.hpp
 * to3dof_canon): the rotation about z by that angle is (r11, r21) / |(r11, r21)| -- no trigonometric call, so the two
 * sides agree to the bit; against the reference's float atan2f / cosf / sinf it differs by a few 1e-8.             */
void sfo_to3dof(float* T) {
  const double r11 = (double)T[0], r21 = (double)T[4];
  const double h = sqrt(r11 * r11 + r21 * r21);
  float c = 1.0f, s = 0.0f;
  if (h > 0.0) { c = (float)(r11 / h); s = (float)(r21 / h); }
  const float x = T[3], y = T[7];
  T[0] = c; T[1] = -s; T[2] = 0.0f; T[3] = x;
  T[4] = s; T[5] = c; T[6] = 0.0f; T[7] = y;
  T[8] = 0.0f; T[9] = 0.0f; T[10] = 1.0f; T[11] = 0.0f;
}

/* Eigen::Quaternionf(rotation) [upstream Eigen Quaternion.h quaternionbase_assign_impl], in double on the float
 * entries; no sign convention applied. */
static void sfo_quat_from_rot(const float* T, double q[4]) {
  double m[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = (double)T[4 * i + j];
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
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
}

/* Transform::inverse() of a rigid transform, in the form the estimators already use for their own result. */
static void sfo_rigid_inverse(const float* T, float* out) {
  double R[9], t[3];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)T[4 * i + j]; t[i] = (double)T[4 * i + 3]; }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out[4 * i + j] = (float)R[3 * j + i];
    out[4 * i + 3] = (float)(-((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]));
  }
}

/* a.interpolate(0.5f, b) [upstream rtabmap Transform::interpolate]: qa.slerp(0.5, qb), translation a + 0.5 (b - a),
 * then Transform(x, y, z, qx, qy, qz, qw), which NORMALISES the quaternion before toRotationMatrix().  At t = 0.5
 * Eigen's slerp weights are equal (sin(theta / 2) / sin(theta) each; 0.5 each for nearly parallel quaternions) with
 * the second negated when qa . qb < 0, so the normalised result is (qa +- qb) / |qa +- qb|: no trigonometry needed. */
void sfo_interpolate_half(const float* A, const float* B, float* out) {
  double qa[4], qb[4], q[4];
  sfo_quat_from_rot(A, qa);
  sfo_quat_from_rot(B, qb);
  const double d = ((qa[0] * qb[0] + qa[1] * qb[1]) + qa[2] * qb[2]) + qa[3] * qb[3];
  for (int i = 0; i < 4; ++i) q[i] = d < 0.0 ? qa[i] - qb[i] : qa[i] + qb[i];
  const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  out[0] = (float)(1.0 - (tyy + tzz)); out[1] = (float)(txy - twz);         out[2] = (float)(txz + twy);
  out[4] = (float)(txy + twz);         out[5] = (float)(1.0 - (txx + tzz)); out[6] = (float)(tyz - twx);
  out[8] = (float)(txz - twy);         out[9] = (float)(tyz + twx);         out[10] = (float)(1.0 - (txx + tyy));
  for (int i = 0; i < 3; ++i) out[4 * i + 3] = A[4 * i + 3] + 0.5f * (B[4 * i + 3] - A[4 * i + 3]);
}

/* One computeTransformationFromFeats call (myRegistration.cpp:225-303 ->
 * myRegistrationVis.cpp:441-1410) with estimation type 0 or 1; forward-only estimation, or (estimation type 0 without
 * bundle adjustment) both directions merged as :1155-1189 and :1376-1394 do.  Reg/Force3DoF: :245-248, :1100-1102 /
 * :1141-1143 and myRegistration.cpp:269-276. */
static int sfo_registration_pass(const sf_params* p, const sf_features* from, const sf_features* to,
                                 const float* guess, int guess_is_null, sfo_pass* out, int* guided,
                                 uint16_t* cf, uint16_t* ct, int* nc) {
  memset(out, 0, sizeof(*out));
  out->is_null = 1;
  out->cov_diag = 1.0;  /* myRegistrationVis.cpp:923 */
  out->cov_diag_ang = 1.0;
  *guided = 0;
  *nc = 0;
  float guess3[12];
  if (!guess_is_null && p->force_3dof) {     /* myRegistration.cpp:245-248 */
    memcpy(guess3, guess, sizeof(guess3));
    sfo_to3dof(guess3);
    guess = guess3;
  }
  int kf = from->rows, kt = to->rows;
  int n_words_from = 0, n_words_to = 0, n_words_to_2d = 0, all_outside = 0, rc;
  if (kf > 0 && kt > 0) {
    if (from->cols != to->cols) return SF_EINVAL;  /* :683 UASSERT */
    int calibrated = p->image_width > 0 && p->image_height > 0 && p->fx > 0.0 && p->fy > 0.0; /* :469-474 */
    int guess_set = !guess_is_null && !sfo_is_identity(guess);                                   /* :477 */
    if (guess_set && p->guess_win_size > 0 && from->n3d > 0 && calibrated) {                     /* :478-479 */
      *guided = 1;
      rc = sfo_match_guided(p, guess, from->desc, from->xyz, from->kpts, kf, to->desc, to->kpts, kt,
                            to->n3d > 0, from->cols, cf, ct, nc, &n_words_from, &n_words_to,
                            &n_words_to_2d, &all_outside);
    } else {
      rc = sfo_match_global(from->desc, kf, to->desc, kt, from->cols, p->nndr, from->n3d > 0,
                            to->n3d > 0, cf, ct, nc, &n_words_from, &n_words_to, &n_words_to_2d, p->desc_type);
    }
    if (rc != SF_OK) return rc;
  } else if (kf > 0) {
    n_words_from = from->n3d > 0 ? kf : 0;   /* :897-910 */
  }
  /* :928 if (wordsTo.size()) ... else "Missing correspondences" */
  if (n_words_to_2d > 0) {
    const int calibrated = p->image_width > 0 && p->image_height > 0 && p->fx > 0.0 && p->fy > 0.0;
    const int pnp = p->estimation_type == 1;
    /* wordsFrom.size(): every "from" row is a word of the global matcher (:856-875, ids distinct by construction); the
     * guided matcher makes a word of every row with a finite 3D point (:766-774, :793-802) */
    const int n_words_from_2d = *guided ? n_words_from : kf;
    /* dir = 0 (:936-960 A = from, B = to): 3D-3D gate :1117-1118 ; PnP gate :1059 (isValidForProjection) and :1070-1071.
     * dir = 1 (:961-977 A = to, B = from; Vis/ForwardEstOnly = false only): the same gates with the roles swapped */
    const int gate[2] = {
        pnp ? (calibrated && n_words_from >= p->min_inliers && n_words_to_2d >= p->min_inliers)
            : (n_words_from >= p->min_inliers && n_words_to >= p->min_inliers),
        p->forward_est_only ? 0
            : pnp ? (calibrated && n_words_to >= p->min_inliers && n_words_from_2d >= p->min_inliers)
                  : (n_words_to >= p->min_inliers && n_words_from >= p->min_inliers)};
    sfo_motion mo[2];
    uint8_t* imask[2] = {NULL, NULL};
    for (int dir = 0; dir < 2; ++dir) {
      memset(&mo[dir], 0, sizeof(mo[dir]));
      mo[dir].is_null = 1;
      mo[dir].variance = 1.0;            /* :934-935 covariances[dir] = eye(6) */
      mo[dir].variance_ang = 1.0;
      imask[dir] = (uint8_t*)calloc((size_t)(*nc > 0 ? *nc : 1), 1);
      if (!imask[dir]) { free(imask[0]); return SF_ENOMEM; }
    }
    rc = SF_OK;
    for (int dir = 0; dir < 2 && rc == SF_OK; ++dir) {
      if (!gate[dir]) continue;
      const sf_features* A = dir == 0 ? from : to;
      const sf_features* B = dir == 0 ? to : from;
      const uint16_t* ca = dir == 0 ? cf : ct;
      const uint16_t* cb = dir == 0 ? ct : cf;
      /* (PnP, dir = 1: the guess -- transforms[0].inverse(), :1087 -- does not enter this restatement's solver, see
       *  sf_oracle_pnp.c; the camera model is the one camera of sf_params for both frames) */
      rc = pnp ? sfo_estimate_motion_3d2d(p, A->xyz, B->kpts, B->n3d > 0 ? B->xyz : NULL, ca, cb, *nc, &mo[dir], imask[dir])
               : sfo_estimate_motion_3d3d(p, A->xyz, B->xyz, ca, cb, *nc, &mo[dir], imask[dir]);
      if (rc == SF_OK && !mo[dir].is_null && p->force_3dof) sfo_to3dof(mo[dir].transform);   /* :1100-1102, :1141-1143 */
    }
    if (rc != SF_OK) { free(imask[0]); free(imask[1]); return rc; }
    if (gate[0] || gate[1]) {
      /* :1155-1189 union of the two directions' inlier and match ids.  3D-3D: both directions match the ids whose two
       * points are finite; PnP: direction d matches the ids whose A-side point is finite. */
      int uni = 0, uni_m = 0;
      for (int i = 0; i < *nc; ++i) {
        uni += (imask[0][i] | imask[1][i]) ? 1 : 0;
        if (pnp) {
          const int m0 = gate[0] && sfo_finite3(from->xyz + 3 * (size_t)cf[i]);
          const int m1 = gate[1] && sfo_finite3(to->xyz + 3 * (size_t)ct[i]);
          uni_m += (m0 | m1) ? 1 : 0;
        }
      }
      out->inliers = uni;
      out->matches = pnp ? uni_m : (mo[0].matches > mo[1].matches ? mo[0].matches : mo[1].matches);
      out->cov_diag = mo[0].variance;
      out->cov_diag_ang = mo[0].variance_ang;
      if (!mo[0].is_null) {
        memcpy(out->transform, mo[0].transform, sizeof(out->transform));
        out->is_null = 0;
      }
      /* :1192-1197 bundle adjustment of the forward transform over the union of the inliers; it ends with
       * transforms[1].setNull() (:1369), so the backward estimate then only contributed its inliers */
      /* (a word that is an inlier of the backward PnP only may have no finite point in the "from" frame; the
       *  reference hands that point to g2o as it is -- [upstream] behaviour undefined -- here such words stay out of the
       *  adjustment and of the inlier count behind it: DESIGN.md section 3) */
      int ba_ran = 0, uni_ba = 0;
      for (int i = 0; i < *nc && from->n3d > 0; ++i)
        uni_ba += ((imask[0][i] | imask[1][i]) && sfo_finite3(from->xyz + 3 * (size_t)cf[i])) ? 1 : 0;
      if (p->bundle_adjustment > 0 && !mo[0].is_null && uni_ba > 0 && n_words_from > 0 && n_words_to_2d > 0) {
        uint8_t* um = imask[0];
        for (int i = 0; i < *nc; ++i)
          um[i] = (uint8_t)((imask[0][i] | imask[1][i]) && sfo_finite3(from->xyz + 3 * (size_t)cf[i]));
        int n_inl = uni_ba, null2 = 0;
        rc = sfo_bundle_adjust(p, from->xyz, from->kpts, to->n3d > 0 ? to->xyz : NULL, to->kpts, cf, ct, um, *nc,
                               out->transform, &n_inl, &null2, NULL);
        if (rc != SF_OK) { free(imask[0]); free(imask[1]); return rc; }
        out->inliers = n_inl;
        out->is_null = null2;
        ba_ran = 1;
      }
      /* :1376-1394 */
      if (!ba_ran && !mo[1].is_null) {
        float inv[12];
        sfo_rigid_inverse(mo[1].transform, inv);
        if (out->is_null) {
          memcpy(out->transform, inv, sizeof(inv));
          out->is_null = 0;
          out->cov_diag = mo[1].variance;
          out->cov_diag_ang = mo[1].variance_ang;
        } else {
          float mid[12];
          sfo_interpolate_half(out->transform, inv, mid);
          memcpy(out->transform, mid, sizeof(mid));
          out->cov_diag = (mo[0].variance + mo[1].variance) / 2.0;
          out->cov_diag_ang = (mo[0].variance_ang + mo[1].variance_ang) / 2.0;
        }
      }
    }
    free(imask[0]); free(imask[1]);
  }
  if (!out->is_null && p->force_3dof) sfo_to3dof(out->transform);   /* myRegistration.cpp:269-276 */
  return SF_OK;
}

/* Eigen's rotation-matrix -> quaternion, then tf::poseEigenToMsg's w >= 0 convention
 * (MsgConversion.cpp:71-81). */
static void sfo_pose_from_transform(const float* T, double pos[3], double q[4]) {
  double m[3][3];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m[i][j] = (double)T[4 * i + j]; pos[i] = (double)T[4 * i + 3]; }
  double x, y, z, w;
  double tr = (m[0][0] + m[1][1]) + m[2][2];
  if (tr > 0.0) {
    double t = sqrt(tr + 1.0);
    w = 0.5 * t; t = 0.5 / t;
    x = (m[2][1] - m[1][2]) * t; y = (m[0][2] - m[2][0]) * t; z = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(((m[i][i] - m[j][j]) - m[k][k]) + 1.0);
    double qv[3];
    qv[i] = 0.5 * t; t = 0.5 / t;
    w = (m[k][j] - m[j][k]) * t;
    qv[j] = (m[j][i] + m[i][j]) * t;
    qv[k] = (m[k][i] + m[i][k]) * t;
    x = qv[0]; y = qv[1]; z = qv[2];
  }
  if (w < 0.0) { x = -x; y = -y; z = -z; w = -w; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

int sfo_estimate_transform_dbg(const sf_params* p, const sf_features* from, const sf_features* to,
                               sf_result* out,
                               uint16_t* c1_from, uint16_t* c1_to, int* n_c1,
                               uint16_t* c2_from, uint16_t* c2_to, int* n_c2) {
  int rc;
  if (!p || !out) return SF_EINVAL;
  if ((rc = sfo_validate(from)) != SF_OK) return rc;
  if ((rc = sfo_validate(to)) != SF_OK) return rc;
  if (p->estimation_type != 0 && p->estimation_type != 1) return SF_EINVAL;
  if (p->desc_type != 0 && p->desc_type != 1) return SF_EINVAL;
  if (p->desc_type == 1 && ((from->rows > 0 && from->cols % 4) || (to->rows > 0 && to->cols % 4))) return SF_EINVAL;
  if (p->estimation_type == 1 && (p->pnp_flags != 0 || p->pnp_refine_iterations < 0)) return SF_EINVAL;
  if (p->bundle_adjustment != 0 &&
      (p->bundle_adjustment != 1 || !(p->image_width > 0 && p->image_height > 0 && p->fx > 0.0 && p->fy > 0.0) ||
       p->ba_iterations < 0 || !(p->ba_pixel_variance > 0.0f) || !(p->ba_robust_kernel_delta > 0.0f) ||
       !(p->stereo_baseline >= 0.0f)))
    return SF_EINVAL;   /* :1230 UASSERT(stereoCameraModelTo.isValidForProjection()) */
  memset(out, 0, sizeof(*out));
  int cap = from->rows > to->rows ? from->rows : to->rows;
  if (cap < 1) cap = 1;
  uint16_t* buf = NULL;
  uint16_t *cf1 = c1_from, *ct1 = c1_to, *cf2 = c2_from, *ct2 = c2_to;
  if (!cf1 || !ct1 || !cf2 || !ct2) {
    buf = (uint16_t*)malloc((size_t)cap * 4 * sizeof(uint16_t));
    if (!buf) return SF_ENOMEM;
    cf1 = buf; ct1 = buf + cap; cf2 = buf + 2 * cap; ct2 = buf + 3 * cap;
  }
  int nc1 = 0, nc2 = 0, g1 = 0, g2 = 0;
  sfo_pass r1, r2;
  static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  /* pass 1: guess = Transform(0,0,0,0,0,0) = identity (stereoCamGeometricTools.cpp:126,141-152) */
  rc = sfo_registration_pass(p, from, to, ident, 0, &r1, &g1, cf1, ct1, &nc1);
  if (rc != SF_OK) { free(buf); return rc; }
  /* pass 2: guess = result (:153-164).  myRegistration.cpp:264 repeatOnce fires only when the
   * guess is null AND the pass succeeded; with deterministic sampling the repeated global pass
   * reproduces the same null result, so it is not re-run here. */
  rc = sfo_registration_pass(p, from, to, r1.transform, r1.is_null, &r2, &g2, cf2, ct2, &nc2);
  if (rc != SF_OK) { free(buf); return rc; }
  if (n_c1) *n_c1 = nc1;
  if (n_c2) *n_c2 = nc2;

  /* myRegistration.cpp:279-295: covariance (never empty here) with diagonal clamped to 1e-9 */
  double cd = r2.cov_diag, ca = r2.cov_diag_ang;
  if (cd <= 1e-9) cd = 1e-9;
  if (ca <= 1e-9) ca = 1e-9;
  for (int i = 0; i < 3; ++i) { out->covariance[7 * i] = cd; out->covariance[7 * (i + 3)] = ca; }
  out->inliers = r2.inliers; out->matches = r2.matches;
  out->inliers_pass1 = r1.inliers; out->matches_pass1 = r1.matches;
  out->pass1_success = (uint8_t)!r1.is_null;
  out->pass2_guided = (uint8_t)g2;
  out->success = (uint8_t)!r2.is_null;                                 /* :168-175 */
  if (!r2.is_null) sfo_pose_from_transform(r2.transform, out->position, out->orientation);
  /* null -> geometry_msgs::Pose() all zeros (MsgConversion.cpp:77-80) */
  free(buf);
  return SF_OK;
}

int sfo_estimate_transform(const sf_params* p, const sf_features* from, const sf_features* to,
                           sf_result* out) {
  return sfo_estimate_transform_dbg(p, from, to, out, NULL, NULL, NULL, NULL, NULL, NULL);
}

void sfo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int sfo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int sfo_estimate_transform_batch(const sf_params* p, const sf_features* from,
                                 const sf_features* to, int n, sf_result* out, int threads) {
  int rc_all = SF_OK;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
#endif
  for (int i = 0; i < n; ++i) {
    int rc = sfo_estimate_transform(p, from + i, to + i, out + i);
    if (rc != SF_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc_all = rc;
    }
  }
  (void)threads;
  return rc_all;
}
