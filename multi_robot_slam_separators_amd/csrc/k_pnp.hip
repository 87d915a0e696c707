// k_pnp.hip -- RANSAC 3D->2D (PnP) motion estimation for one candidate pair per 256-thread workgroup.
//
// Replaces util3d::estimateMotion3DTo2D as called at myRegistrationVis.cpp:1077-1091 of the reference
// (the `_estimationType == 1` branch, :1055-1112) [upstream rtabmap util3d_motion_estimation.cpp ->
// cv::solvePnPRansac: RANSACPointSetRegistrator, squared reprojection error <= reprojError^2,
// RANSACUpdateNumIters with confidence 0.99, final iterative solve on the inliers; then rtabmap's
// frame change (localTransform * pnp).inverse() and its quartile-based covariance], for the
// correspondences produced by k_match / k_guided.  What is and is not OpenCV's arithmetic is listed in
// DESIGN.md (P3P minimal solver on 4-point samples, keyed sampler, cheirality, LM parametrisation).
//
// CDNA4 mapping (same shape as k_ransac):
//   * correspondences with a finite "from" point are compacted once into LDS: world point (float4),
//     pixel offsets from the principal point (float2), and the packed index pair for the late
//     covariance gather;
//   * hypotheses in rounds of 64, one LANE each: wavefront 0 draws the keyed 4-sample and solves P3P
//     in fp64 (Grunert's quartic by Ferrari + bracketed Newton, frames of the two triangles, the
//     4th point picks the root) and parks the 64 models in LDS; then all four wavefronts count
//     inliers, each over a quarter of the points, with LDS BROADCAST reads and a division-free test
//     (A^2 + B^2 <= thr^2 Z^2, Z > 0: 9 + 2 fma, 4 mul, 2 compares per point);
//   * OpenCV's sequential "best so far / update niters" rule is replayed by thread 0 after every
//     round of 64 hypotheses, so later rounds are skipped exactly when the sequential loop would
//     have stopped;
//   * Levenberg-Marquardt on the inliers: per-lane partial normal equations (21 + 6 + 1 doubles),
//     block-wide reduction in the FIXED order of DESIGN.md section 4, then every lane solves the same
//     6x6 system redundantly (identical arithmetic, no broadcast);
//   * order statistics (first quartile of the 3D errors / angles) by rank counting over +inf padded
//     LDS arrays.
// Compiled with -ffp-contract=off.
#include "sf_device_math.hpp"
#include "sf_internal.hpp"
#include "sf_pnp_math.hpp"
#include "k_ba.hip"

namespace {

#define PNP_NSUM 28

// bear: the observations' unit bearings are kept in LDS (the 256-thread form; 24 B per correspondence) -- the narrow
// chains recompute the three a hypothesis needs (the same operations on the same inputs: the same bits) and fit
// five to a CU instead of four
// The per-iteration inlier counts live one ROUND (64 iterations) at a time: the replay consumes a round's counts before
// the next round overwrites them.
// nw: wavefronts of the chain (one row of partial counts each + the valid flags: 27 264 B at K = 500 on two wavefronts --
// the largest working set that still lets SIX chains share a CU's 160 KB, tools/ubench/lds_granule.hip)
__host__ __device__ inline size_t sf_pnp_lds_bytes_dev(int kcap, int iterations, bool bear = true, int nw = 4) {
  (void)iterations;
  return (size_t)kcap * (16 + (bear ? 24 : 0) + 8 + 4 + 4 + 4 + 2) + 128 * 8 + 64 * 4 + 16 * 4 + 12 * 64 * 4 +
         (size_t)(nw + 1) * 64 * 4 + 16 * 4 + 2 * 32 * 8;
}

struct PnpLds {
  float4* obj;      // [kcap] world ("from" base frame) point
  float2* img;      // [kcap] pixel - principal point of the "to" keypoint
  double* bear;     // [kcap][3] unit bearing vector of the "to" keypoint (fp64, computed once per point)
  uint32_t* cidx;   // [kcap] packed (to << 16 | from) feature indices
  float* e1;        // [kcap] squared 3D error of the members (+inf elsewhere)
  float* e2;        // [kcap] angular error
  uint8_t* mask;    // [kcap]
  uint8_t* mask_b;  // [kcap] second inlier set of the refinement rounds
  int* counts;      // [64] inlier counts of the current round's iterations
  double* red;      // [4][32]
  int* misc;        // [16]
  float* hyp;       // [12][64] models of the current round of hypotheses
  int* hyp_cnt;     // [NW][64] partial inlier counts (one row per wavefront) + [64] valid flags
  float* best;      // [12] model of the best hypothesis so far (copied out of `hyp` by the replay)
  double* ne_a;     // [32] normal equations of the current pose   (totals live in LDS, not in VGPRs)
  double* ne_b;     // [32] normal equations of the candidate pose
};

template <int NW = 4>
__device__ __forceinline__ int block_sum_i(int v, int* misc, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (lane == 0) misc[8 + wave] = v;
  __syncthreads();
  int t = misc[8];                          // (integers: the order of the fold is free)
#pragma unroll
  for (int w = 1; w < NW; ++w) t += misc[8 + w];
  return t;
}

struct PnpCam {
  double fx, fy;
  float fxf, fyf, thr2f;
};

// division-free reprojection test of point i under float coefficients (canonical fma chain)
__device__ __forceinline__ bool pnp_inlier(const PnpLds& L, const PnpCam& cam, const float (&c)[12], int i) {
  const float4 P = L.obj[i];
  const float2 o = L.img[i];
  const float X = __fmaf_rn(c[2], P.z, __fmaf_rn(c[1], P.y, __fmaf_rn(c[0], P.x, c[3])));
  const float Y = __fmaf_rn(c[6], P.z, __fmaf_rn(c[5], P.y, __fmaf_rn(c[4], P.x, c[7])));
  const float Z = __fmaf_rn(c[10], P.z, __fmaf_rn(c[9], P.y, __fmaf_rn(c[8], P.x, c[11])));
  const float A = __fmaf_rn(-o.x, Z, cam.fxf * X);
  const float B = __fmaf_rn(-o.y, Z, cam.fyf * Y);
  const float lhs = __fmaf_rn(B, B, A * A);
  // (the threshold through a vector register: the loop vectoriser tests two points per lane with packed-f32
  //  instructions, and as a uniform the threshold pair sat in SGPRs restored by v_readlane right in front of the
  //  packed multiply in k_chain_pnp<8, true> -- the shape tools/pk_isa_scan.py gates on)
  float thr2 = cam.thr2f;
#ifndef SF_NO_PK_BARRIERS
  asm volatile("" : "+v"(thr2));
#endif
  const float rhs = thr2 * (Z * Z);
  return (Z > 0.0f) && (lhs <= rhs);
}

// model of RANSAC iteration `it`: keyed 4-sample, P3P on the first three, the 4th picks the root
template <bool BEAR>
__device__ inline bool pnp_hypothesis(const PnpLds& L, const PnpCam& cam, uint64_t seed, uint32_t it, uint32_t m,
                                      float (&coef)[12]) {
  uint32_t s0, s1, s2, s3;
  sfd::sample_quad(seed, it, 0u, m, s0, s1, s2, s3);
  double P[3][3], f[3][3];
  const uint32_t sk[3] = {s0, s1, s2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 p = L.obj[sk[k]];
    P[k][0] = (double)p.x; P[k][1] = (double)p.y; P[k][2] = (double)p.z;
    if constexpr (BEAR) {
      const double* b = L.bear + 3 * sk[k];
      f[k][0] = b[0]; f[k][1] = b[1]; f[k][2] = b[2];
    } else {                               // (the operations of the precomputation in pnp_body, on the same inputs)
      const float2 o = L.img[sk[k]];
      const double un = (double)o.x / cam.fx, vn = (double)o.y / cam.fy;
      const double inv = 1.0 / sqrt((un * un + vn * vn) + 1.0);
      f[k][0] = un * inv; f[k][1] = vn * inv; f[k][2] = inv;
    }
  }
  const float4 p4 = L.obj[s3];
  const float2 o4 = L.img[s3];
  const double P4[3] = {(double)p4.x, (double)p4.y, (double)p4.z};
  return sfd::p3p_best(P, f, P4, (double)o4.x, (double)o4.y, cam.fx, cam.fy, coef);
}

// Normal equations of the reprojection error over the members of L.mask at pose (q, t).
template <int NW>
__device__ inline void pnp_normal_eq(const PnpLds& L, const PnpCam& cam, int m, const uint8_t* mask,
                                     const double (&q)[4], const double (&t)[3], double* out, int tid) {
  double R[9];
  sfd::quat_to_R(q, R);
  sfd::canon_reduce_to_lds<PNP_NSUM, 32, NW>(m, tid, L.red, out, [&](int i, double (&ne)[PNP_NSUM]) {
    if (!mask[i]) return;
    const float4 p = L.obj[i];
    const float2 o = L.img[i];
    const double Px = (double)p.x, Py = (double)p.y, Pz = (double)p.z;
    const double Yx = (R[0] * Px + R[1] * Py) + R[2] * Pz;
    const double Yy = (R[3] * Px + R[4] * Py) + R[5] * Pz;
    const double Yz = (R[6] * Px + R[7] * Py) + R[8] * Pz;
    const double X = Yx + t[0], Y = Yy + t[1], Z = Yz + t[2];
    if (Z > 0.0) {
      const double iz = 1.0 / Z;
      const double xn = X * iz, yn = Y * iz;
      const double ru = cam.fx * xn - (double)o.x;
      const double rv = cam.fy * yn - (double)o.y;
      const double a0 = cam.fx * iz, a2 = -(a0 * xn);
      const double b1 = cam.fy * iz, b2 = -(b1 * yn);
      double Ju[6], Jv[6];
      Ju[0] = a2 * Yy;            Ju[1] = a0 * Yz - a2 * Yx; Ju[2] = -(a0 * Yy);
      Ju[3] = a0;                 Ju[4] = 0.0;               Ju[5] = a2;
      Jv[0] = b2 * Yy - b1 * Yz;  Jv[1] = -(b2 * Yx);        Jv[2] = b1 * Yx;
      Jv[3] = 0.0;                Jv[4] = b1;                Jv[5] = b2;
      {
        int o_ = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int k = j; k < 6; ++k) { ne[o_] += Ju[j] * Ju[k] + Jv[j] * Jv[k]; ++o_; }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) ne[21 + j] += Ju[j] * ru + Jv[j] * rv;
      ne[27] += ru * ru + rv * rv;
    } else {
      ne[27] += 1e30;   // a member behind the camera makes the pose unacceptable
    }
  });
}

// Levenberg-Marquardt over the members of `mask` from pose (q, t): at most 20 evaluations, diagonal
// scaled by 1 + lambda [upstream cvFindExtrinsicCameraParams2 / CvLevMarq].  Every lane runs the same
// scalar control flow on the block-reduced sums.  ne returns the normal equations at the final pose.
template <int NW>
__device__ inline void pnp_lm(const PnpLds& L, const PnpCam& cam, int m, const uint8_t* mask, double (&q)[4],
                              double (&t)[3], int tid) {
  // L.ne_a holds the normal equations of the accepted pose on return
  double* cur = L.ne_a;
  double* cand = L.ne_b;
  pnp_normal_eq<NW>(L, cam, m, mask, q, t, cur, tid);
  double lambda = 1e-3;
  // The damped 6 x 6 solve and the candidate pose are a short scalar computation on the reduced sums: ONE wavefront
  // does it and hands the candidate over through L.red (free between two reductions); round 2 had all four wavefronts
  // issue the same ~300 fp64 instructions per evaluation, on all four SIMDs of the CU the chain shares with two others.
  const int lm_wave = tid >> 6, lm_lane = tid & 63;
  for (int iter = 0; iter < 20; ++iter) {
    if (lm_wave == 0) {
      double d[6];
      const bool ok = sfd::solve6(cur, lambda, d);
      double qc0[4], tc0[3], dd0 = 0.0, tt0 = 0.0;
      if (ok) {
        const double hx = 0.5 * d[0], hy = 0.5 * d[1], hz = 0.5 * d[2];
        const double dn = 1.0 / sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
        const double dw = dn, dx = hx * dn, dy = hy * dn, dz = hz * dn;
        qc0[0] = ((dw * q[0] - dx * q[1]) - dy * q[2]) - dz * q[3];
        qc0[1] = ((dw * q[1] + dx * q[0]) + dy * q[3]) - dz * q[2];
        qc0[2] = ((dw * q[2] - dx * q[3]) + dy * q[0]) + dz * q[1];
        qc0[3] = ((dw * q[3] + dx * q[2]) - dy * q[1]) + dz * q[0];
        const double qn = 1.0 / sqrt(((qc0[0] * qc0[0] + qc0[1] * qc0[1]) + qc0[2] * qc0[2]) + qc0[3] * qc0[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) qc0[i] = qc0[i] * qn;
#pragma unroll
        for (int i = 0; i < 3; ++i) tc0[i] = t[i] + d[3 + i];
        // a step below float epsilon relative to the parameters ends the iteration either way
        dd0 = ((((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]) + d[4] * d[4]) + d[5] * d[5];
        tt0 = ((tc0[0] * tc0[0] + tc0[1] * tc0[1]) + tc0[2] * tc0[2]) + 1.0;
      }
      if (lm_lane == 0) {
        L.red[0] = ok ? 1.0 : 0.0;
        if (ok) {
          L.red[1] = qc0[0]; L.red[2] = qc0[1]; L.red[3] = qc0[2]; L.red[4] = qc0[3];
          L.red[5] = tc0[0]; L.red[6] = tc0[1]; L.red[7] = tc0[2];
          L.red[8] = dd0; L.red[9] = tt0;
        }
      }
    }
    __syncthreads();
    if (L.red[0] == 0.0) {                       // (uniform: the factorisation failed at this damping)
      __syncthreads();                           // every wavefront has read the flag before the next hand-over
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
      continue;
    }
    double qc[4], tc[3];
    qc[0] = L.red[1]; qc[1] = L.red[2]; qc[2] = L.red[3]; qc[3] = L.red[4];
    tc[0] = L.red[5]; tc[1] = L.red[6]; tc[2] = L.red[7];
    const double dd = L.red[8], tt = L.red[9];
    pnp_normal_eq<NW>(L, cam, m, mask, qc, tc, cand, tid);     // (its first barrier: the hand-over has been read by everyone)
    if (cand[27] < cur[27]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = qc[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = tc[i];
      double* sw = cur; cur = cand; cand = sw;
      lambda = lambda * 0.1;
      if (lambda < 1e-16) lambda = 1e-16;
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e12) break;
    }
    if (dd <= 1.4e-14 * tt) break;
  }
  if (cur != L.ne_a) {   // leave the accepted pose's sums in ne_a (uniform branch)
    __syncthreads();
    if (tid < PNP_NSUM) L.ne_a[tid] = cur[tid];
    __syncthreads();
  }
}

// rtabmap computeReprojErrors: members = points in front of the camera with reprojection error (px,
// not squared) <= thr under the float-rounded pose; L.e1[i] receives member i's error.
template <int NW>
__device__ inline int pnp_select(const PnpLds& L, const PnpCam& cam, int m, const double (&q)[4],
                                 const double (&t)[3], float thr, uint8_t* mask, int tid) {
  double Rd[9];
  sfd::quat_to_R(q, Rd);
  float c[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c[4 * i + j] = (float)Rd[3 * i + j];
    c[4 * i + 3] = (float)t[i];
  }
  int n = 0;
  for (int i = tid; i < m; i += 64 * NW) {
    const float4 P = L.obj[i];
    const float2 o = L.img[i];
    const float X = __fmaf_rn(c[2], P.z, __fmaf_rn(c[1], P.y, __fmaf_rn(c[0], P.x, c[3])));
    const float Y = __fmaf_rn(c[6], P.z, __fmaf_rn(c[5], P.y, __fmaf_rn(c[4], P.x, c[7])));
    const float Z = __fmaf_rn(c[10], P.z, __fmaf_rn(c[9], P.y, __fmaf_rn(c[8], P.x, c[11])));
    bool in = false;
    if (Z > 0.0f) {
      const float du = __fmaf_rn(cam.fxf, (X / Z), -o.x);
      const float dv = __fmaf_rn(cam.fyf, (Y / Z), -o.y);
      const float e = sqrtf(__fmaf_rn(dv, dv, du * du));
      in = e <= thr;
      if (in) L.e1[i] = e;
    }
    mask[i] = in ? 1 : 0;
    n += in ? 1 : 0;
  }
  return block_sum_i<NW>(n, L.misc, tid);
}

// value of rank `rank` among the finite entries of arr[0..m4) (+inf padded): rank counting, 4 per read
template <int NT>
__device__ inline float rank_value(const float* arr, int m, int rank, double* slot, int tid) {
  const int m4 = (m + 3) & ~3;
  __syncthreads();
  for (int i = tid; i < m; i += NT) {
    const float v = arr[i];
    if (v < __int_as_float(0x7F800000)) {
      int lt = 0, eq = 0;
      const float4* d4 = reinterpret_cast<const float4*>(arr);
      for (int j = 0; j < m4 / 4; ++j) {
        const float4 u = d4[j];
        lt += (u.x < v ? 1 : 0) + (u.y < v ? 1 : 0) + (u.z < v ? 1 : 0) + (u.w < v ? 1 : 0);
        eq += (u.x == v ? 1 : 0) + (u.y == v ? 1 : 0) + (u.z == v ? 1 : 0) + (u.w == v ? 1 : 0);
      }
      if (lt <= rank && rank < lt + eq) *slot = (double)v;  // every writer holds the same value
    }
  }
  __syncthreads();
  return (float)*slot;
}

// What a finished pass leaves in LDS for the bundle adjustment (k_ba.hip): the compacted "from" points, the packed
// feature indices, the final inlier mask.  ran = false: the pass ended early (no estimate to refine).
struct PnpTail {
  const float4* obj;
  const uint32_t* cidx;
  const uint8_t* inl;
  int m;
  bool ran;
};

// Body of one PnP pass for ONE pair (the calling 256-thread workgroup); smem_raw is the workgroup's dynamic LDS.
// DIR = 1: the backward estimate of Vis/ForwardEstOnly = false (myRegistrationVis.cpp:961-977: A = "to", B = "from" --
// the 3D points of "to" against the keypoints of "from"; the stage kernel only).  `mask_out` (stage kernel, both
// directions): one byte per "from" feature of the pair, set for this estimate's inliers; `guided` (with mask_out): the
// pass's correspondences come from guided matching (decides what the 2D words of "from" are, sf_pnp_dir_gate).
template <int DIR = 0, int NW = 4>
__device__ __forceinline__ PnpTail pnp_body(const StoreView& st, int pair, const int32_t* __restrict__ pair_from,
                                         const int32_t* __restrict__ pair_to, const uint32_t* __restrict__ corr,
                                         const CorrHeader* __restrict__ hdr, PassState* __restrict__ pass,
                                         const DeviceParams& P, unsigned char* smem_raw, int trace_base = 0,
                                         uint8_t* mask_out = nullptr, bool guided = false) {
  // (trace_base: first of five timestamp slots of the diagnostic build -- gather, RANSAC, first solve, refinement
  //  rounds, pose + covariance; SF_TRACE_MARK compiles to nothing in the product)
  constexpr int NT = 64 * NW;
  constexpr bool BEAR = NW == 4;          // (sf_pnp_lds_bytes_dev)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  const PnpTail none = {nullptr, nullptr, nullptr, 0, false};
  (void)trace_base;
  const int max_it = P.iterations > 0 ? P.iterations : 0;
  const bool to_has_3d = st.meta[DIR ? sF : sT].y > 0;   // the B frame carries 3D points (selects the covariance form)

  PnpLds L;
  {
    unsigned char* p = smem_raw;
    L.obj = (float4*)p; p += (size_t)kcap * 16;
    L.red = (double*)p; p += 128 * 8;
    L.bear = (double*)p; p += BEAR ? (size_t)kcap * 24 : 0;
    L.img = (float2*)p; p += (size_t)kcap * 8;
    L.e1 = (float*)p; p += (size_t)kcap * 4;
    L.e2 = (float*)p; p += (size_t)kcap * 4;
    L.cidx = (uint32_t*)p; p += (size_t)kcap * 4;
    L.counts = (int*)p; p += 64 * 4;
    L.misc = (int*)p; p += 16 * 4;
    L.mask = p; p += kcap;
    L.mask_b = p; p += kcap;
    L.hyp = (float*)p; p += 12 * 64 * 4;
    L.hyp_cnt = (int*)p; p += (NW + 1) * 64 * 4;
    L.best = (float*)p; p += 16 * 4;
    L.ne_a = (double*)p; p += 32 * 8;
    L.ne_b = (double*)p;
  }

  // ---- estimateMotion3DTo2D: ids of words2B found in words3A with a finite 3D point ---------------
  const int n_corr = hdr[pair].n_corr;
  const uint32_t* cl = corr + (size_t)pair * kcap;
  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  const float4* kT = st.kp + (size_t)(DIR ? sF : sT) * kcap;      // (keypoints of the B frame)
  const float cxf = (float)P.cx, cyf = (float)P.cy;
  if (P.bidirectional && !sf_pnp_dir_gate(DIR, hdr[pair], st.meta[sF].x, guided, P.min_inliers)) {
    // this direction's gate (:1059, :1070-1071) is closed: covariances[dir] stays the identity, no matches, no inliers
    if (tid == 0) {
      PassState z;
#pragma unroll
      for (int i = 0; i < 12; ++i) z.T[i] = 0.f;
      z.var = 1.0; z.var_ang = 1.0; z.is_null = 1; z.inliers = 0; z.matches = 0; z.pad = 0;
      pass[pair] = z;
    }
    return none;
  }
  if (tid < 16) L.misc[tid] = 0;
  __syncthreads();
  int m = 0;
  for (int base = 0; base < n_corr; base += NT) {
    const int i = base + tid;
    bool ok = false;
    float ax = 0, ay = 0, az = 0, ox = 0, oy = 0;
    uint32_t c = 0;
    if (i < n_corr) {
      c = cl[i];
      const float* a = DIR ? xT + 3 * (c >> 16) : xF + 3 * (c & 0xFFFFu);
      ax = a[0]; ay = a[1]; az = a[2];
      const float4 kp = kT[DIR ? (c & 0xFFFFu) : (c >> 16)];
      ox = kp.x - cxf; oy = kp.y - cyf;
      ok = sfd::finite3(ax, ay, az);
    }
    const unsigned long long bal = __ballot(ok);
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
    if (ok) {
      L.obj[m + woff + before] = make_float4(ax, ay, az, 0.f);
      L.img[m + woff + before] = make_float2(ox, oy);
      L.cidx[m + woff + before] = c;
    }
    m += total;
    __syncthreads();
  }

  // unit bearings of the observations, once per point (the P3P of every hypothesis reads three of them)
  if constexpr (BEAR) {
    for (int i = tid; i < m; i += NT) {
      const float2 o = L.img[i];
      const double un = (double)o.x / P.fx, vn = (double)o.y / P.fy;
      const double inv = 1.0 / sqrt((un * un + vn * vn) + 1.0);
      L.bear[3 * i] = un * inv; L.bear[3 * i + 1] = vn * inv; L.bear[3 * i + 2] = inv;
    }
  }
  __syncthreads();

  PassState ps;
#pragma unroll
  for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
  ps.var = 1.0;
  ps.var_ang = 1.0;
  ps.is_null = 1;
  ps.inliers = 0;
  ps.matches = m;
  ps.pad = 0;
  if (m < P.min_inliers || m < 4) {
    if (tid == 0) pass[pair] = ps;
    return none;
  }

  if (P.dbg_stop == 1) { if (tid == 0) pass[pair] = ps; return none; }   // diagnostic truncation (SF_RANSAC_STOP)
  if (trace_base) SF_TRACE_MARK(P, pair, trace_base);
  PnpCam cam;
  cam.fx = P.fx; cam.fy = P.fy;
  cam.fxf = (float)P.fx; cam.fyf = (float)P.fy;
  cam.thr2f = P.pnp_thr2f;

  // ---- cv::RANSACPointSetRegistrator::run: one lane per hypothesis, rounds of SF_BLOCK ---------------
  {
    int niters = max_it, best = 0, best_it = -1, sc_it = 0;   // thread 0 only
    if (tid == 0) { L.misc[0] = -1; L.misc[1] = 1; }
    const int slice = (m + NW - 1) / NW;                   // points counted by each wavefront
    const int i0 = min(m, wave * slice), i1 = min(m, i0 + slice);
    int* hv = L.hyp_cnt + NW * 64;
    const int fit_wave = pair & (NW - 1);   // rotate the solving wavefront (SIMD) with the pair
    for (int base = 0; base < max_it; base += 64) {
      const int it = base + lane;
      if (wave == fit_wave) {
        int valid = 0;
        if (it < max_it) {
          float coef[12];
          if (pnp_hypothesis<BEAR>(L, cam, P.seed, (uint32_t)it, (uint32_t)m, coef)) {
#pragma unroll
            for (int k = 0; k < 12; ++k) L.hyp[k * 64 + lane] = coef[k];
            valid = 1;
          }
        }
        hv[lane] = valid;
      }
      __syncthreads();
      {
        int cnt = 0;
        if (hv[lane]) {
          float coef[12];
#pragma unroll
          for (int k = 0; k < 12; ++k) coef[k] = L.hyp[k * 64 + lane];
          for (int i = i0; i < i1; ++i) cnt += pnp_inlier(L, cam, coef, i) ? 1 : 0;   // LDS broadcast reads
        }
        L.hyp_cnt[wave * 64 + lane] = cnt;
      }
      __syncthreads();
      if (wave == fit_wave && it < max_it) {
        int tot = L.hyp_cnt[lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += L.hyp_cnt[64 * w + lane];
        L.counts[lane] = tot;
      }
      __syncthreads();
      if (tid == 0) {
        const int lim = min(max_it, base + 64);   // counts exist for iterations < lim
        while (sc_it < niters && sc_it < lim) {
          const int good = L.counts[sc_it - base];
          const int bar = best > 3 ? best : 3;          // max(maxGoodCount, modelPoints - 1)
          if (good > bar) {
            best = good;
            best_it = sc_it;
#pragma unroll
            for (int k = 0; k < 12; ++k) L.best[k] = L.hyp[k * 64 + (sc_it - base)];   // keep the winning model
            if (P.adaptive_stop) niters = sfd::update_num_iters(0.99, (double)(m - good) / (double)m, 4, niters);
          }
          ++sc_it;
        }
        L.misc[0] = best_it;
        L.misc[1] = (sc_it >= niters) ? 1 : 0;
      }
      __syncthreads();
      if (L.misc[1]) break;
    }
    __syncthreads();
  }
  if (P.dbg_stop == 2) { if (tid == 0) pass[pair] = ps; return none; }
  if (trace_base) SF_TRACE_MARK(P, pair, trace_base + 1);
  const int best_it = L.misc[0];
  if (best_it < 0) {   // solvePnPRansac returned false: no inliers
    if (tid == 0) pass[pair] = ps;
    return none;
  }

  // ---- winning model (parked in LDS by the replay) and its inlier mask ---------------------------------
  float coef[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) coef[k] = L.best[k];
  int n_inl = 0;
  for (int i = tid; i < m; i += NT) {
    const bool in = pnp_inlier(L, cam, coef, i);
    L.mask[i] = in ? 1 : 0;
    n_inl += in ? 1 : 0;
  }
  n_inl = block_sum_i<NW>(n_inl, L.misc, tid);   // also orders the mask writes before the reads below

  if (P.dbg_stop == 3) { if (tid == 0) pass[pair] = ps; return none; }
  // ---- final solve on the inliers: Levenberg-Marquardt, at most 20 evaluations ------------------------
  double q[4], t[3];
  {
    double Rb[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rb[3 * i + j] = (double)coef[4 * i + j];
      t[i] = (double)coef[4 * i + 3];
    }
    sfd::R_to_quat(Rb, q);
  }
  pnp_lm<NW>(L, cam, m, L.mask, q, t, tid);
  if (trace_base) SF_TRACE_MARK(P, pair, trace_base + 2);

  // ---- rtabmap util3d::solvePnPRansac refinement rounds (Vis/PnPRefineIterations > 0) ---------------------
  const uint8_t* inl = L.mask;
  const int min_count = P.min_inliers > 4 ? P.min_inliers : 4;
  if (n_inl >= min_count && P.pnp_refine_iterations > 0) {
    const float inlier_thr = P.pnp_reproj_error;
    float error_threshold = inlier_thr;
    int refine_iterations = 0;
    bool inlier_changed = false;
    uint8_t* prev = L.mask;
    uint8_t* neu = L.mask_b;
    int n_prev = n_inl, n_new = 0;
    for (int i = tid; i < m; i += NT) neu[i] = 0;
    int n_sizes = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0;   // last four pushed sizes (z1 newest)
    do {
      pnp_lm<NW>(L, cam, m, prev, q, t, tid);                // solvePnP from the current model
      z4 = z3; z3 = z2; z2 = z1; z1 = n_prev;
      ++n_sizes;
      __syncthreads();
      n_new = pnp_select<NW>(L, cam, m, q, t, error_threshold, neu, tid);
      if (n_new < min_count) {
        ++refine_iterations;
        if (refine_iterations >= P.pnp_refine_iterations) break;
        continue;
      }
      // uMean / uVariance of the members' errors (block-order sums, float results)
      double s1[1] = {0.0};
      if constexpr (NW == 4) {
        for (int i = tid; i < m; i += NT) if (neu[i]) s1[0] += (double)L.e1[i];
        sfd::block_sum_canon<1, 32>(s1, L.red, tid);
      } else {
        sfd::canon_reduce<1, 32, NW>(m, tid, L.red, s1, [&](int i, double (&a)[1]) { if (neu[i]) a[0] += (double)L.e1[i]; });
      }
      const float mean = (float)(s1[0] / (double)n_new);
      float variance = 0.0f;
      if (n_new > 1) {
        double s2[1] = {0.0};
        if constexpr (NW == 4) {
          for (int i = tid; i < m; i += NT) {
            if (neu[i]) { const float dlt = L.e1[i] - mean; s2[0] += (double)(dlt * dlt); }
          }
          sfd::block_sum_canon<1, 32>(s2, L.red, tid);
        } else {
          sfd::canon_reduce<1, 32, NW>(m, tid, L.red, s2, [&](int i, double (&a)[1]) {
            if (neu[i]) { const float dlt = L.e1[i] - mean; a[0] += (double)(dlt * dlt); }
          });
        }
        variance = (float)(s2[0] / (double)(n_new - 1));
      }
      const float sthr = (float)P.refine_sigma * sqrtf(variance);
      error_threshold = sthr < inlier_thr ? sthr : inlier_thr;
      inlier_changed = false;
      { uint8_t* tp = prev; prev = neu; neu = tp; const int tn = n_prev; n_prev = n_new; n_new = tn; }
      if (n_new != n_prev) {
        if (n_sizes >= 4 && z1 == z3 && z2 == z4) break;   // oscillating
        inlier_changed = true;
        continue;
      }
      int diff = 0;
      for (int i = tid; i < m; i += NT) diff |= (prev[i] != neu[i]) ? 1 : 0;
      inlier_changed = block_sum_i<NW>(diff, L.misc, tid) != 0;
    } while (inlier_changed && ++refine_iterations < P.pnp_refine_iterations);
    inl = neu;
    n_inl = n_new;
  }

  if (P.dbg_stop == 4) { if (tid == 0) pass[pair] = ps; return none; }
  if (trace_base) SF_TRACE_MARK(P, pair, trace_base + 3);
  if (mask_out)
    for (int i = tid; i < m; i += NT)
      if (inl[i]) mask_out[L.cidx[i] & 0xFFFFu] = 1;
  ps.inliers = n_inl;
  if (n_inl < P.min_inliers) {
    if (tid == 0) pass[pair] = ps;
    return none;
  }

  // ---- transform = (localTransform * pnp).inverse()  (rtabmap::Transform is float) ----------------------
  float T[12];
  {
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
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) T[4 * i + j] = MR[3 * j + i];
      T[4 * i + 3] = -((MR[i] * Mt[0] + MR[3 + i] * Mt[1]) + MR[6 + i] * Mt[2]);
    }
  }
  bool allz = true;
#pragma unroll
  for (int i = 0; i < 12; ++i) { ps.T[i] = T[i]; allz = allz && (T[i] == 0.f); }
  ps.is_null = allz ? 1 : 0;

  // ---- covariance [upstream estimateMotion3DTo2D] ---------------------------------------------------------
  if (to_has_3d) {
    const int m4 = (m + 3) & ~3;
    int cnt = 0;
    for (int i = tid; i < m4; i += NT) {
      float v1 = __int_as_float(0x7F800000), v2 = __int_as_float(0x7F800000);
      if (i < m && inl[i]) {
        const float* b = DIR ? xF + 3 * (L.cidx[i] & 0xFFFFu) : xT + 3 * (L.cidx[i] >> 16);
        const float bx = b[0], by = b[1], bz = b[2];
        if (sfd::finite3(bx, by, bz)) {
          const float4 a = L.obj[i];
          const float nx = __fmaf_rn(T[2], bz, __fmaf_rn(T[1], by, __fmaf_rn(T[0], bx, T[3])));
          const float ny = __fmaf_rn(T[6], bz, __fmaf_rn(T[5], by, __fmaf_rn(T[4], bx, T[7])));
          const float nz = __fmaf_rn(T[10], bz, __fmaf_rn(T[9], by, __fmaf_rn(T[8], bx, T[11])));
          const float dx = nx - a.x, dy = ny - a.y, dz = nz - a.z;
          v1 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
          const double u1[3] = {(double)(a.x - T[3]), (double)(a.y - T[7]), (double)(a.z - T[11])};
          const double u2[3] = {(double)(nx - T[3]), (double)(ny - T[7]), (double)(nz - T[11])};
          const double cr[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2],
                                u1[0] * u2[1] - u1[1] * u2[0]};
          v2 = (float)sfd::canon_atan2(sqrt(sfd::dot3(cr, cr)), sfd::dot3(u1, u2));
          ++cnt;
        }
      }
      L.e1[i] = v1;
      L.e2[i] = v2;
    }
    const int oi = block_sum_i<NW>(cnt, L.misc, tid);
    if (oi > 0) {
      ps.var = 2.1981 * (double)rank_value<NT>(L.e1, m, oi >> 2, &L.red[120], tid);
      ps.var_ang = 2.1981 * (double)rank_value<NT>(L.e2, m, oi >> 2, &L.red[121], tid);
    }
  } else {
    pnp_normal_eq<NW>(L, cam, m, inl, q, t, L.ne_a, tid);
    // sqrtf and operator/ are IEEE-exact on gfx950; __fsqrt_rn is NOT (native v_sqrt_f32, ~1 ulp:
    // tools/ubench/fp_case.hip) and must not appear in canonical arithmetic
    const double v = (double)sqrtf((float)L.ne_a[27] / (float)n_inl);
    ps.var = v;
    ps.var_ang = v;
  }
  if (tid == 0) {
    if (P.force_3dof && !ps.is_null) sfd::to3dof_canon(ps.T);     // myRegistrationVis.cpp:1100-1102
    pass[pair] = ps;
  }
  if (trace_base) SF_TRACE_MARK(P, pair, trace_base + 4);
  return PnpTail{L.obj, L.cidx, inl, m, true};
}

template <int DIR = 0>
__global__ void __launch_bounds__(SF_BLOCK, 3)
k_pnp(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
      const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
      const uint32_t* __restrict__ corr, const CorrHeader* __restrict__ hdr,
      PassState* __restrict__ pass, int extra_3dof, DeviceParams P, uint8_t* __restrict__ mask = nullptr,
      const uint8_t* __restrict__ guided_flag = nullptr) {
  if ((int)blockIdx.x >= *counter) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = list[blockIdx.x];
  const PnpTail tail = pnp_body<DIR>(st, pair, pair_from, pair_to, corr, hdr, pass, P, smem_raw, 0,
                                     mask ? mask + (size_t)pair * st.kcap : nullptr,
                                     guided_flag != nullptr && guided_flag[pair] != 0);
  (void)tail;      // (myRegistrationVis.cpp:1192-1370, the adjustment of this estimate: k_ba_pass, its own launch)
  // myRegistration.cpp:269-276, and for pass 1 the application its result meets as the guess of pass 2 (:245-248)
  if (extra_3dof && threadIdx.x == 0 && !pass[pair].is_null)
    for (int t = 0; t < extra_3dof; ++t) sfd::to3dof_canon(pass[pair].T);
}

// Vis/ForwardEstOnly = false with the PnP estimator: the two directions' estimates of a pass merged as
// myRegistrationVis.cpp:1155-1189 (union of the inlier ids and of the match ids -- direction d matches the ids whose
// A-side point is finite, if its gate was open) and :1376-1394 (inverse of the backward transform; interpolate(0.5) when
// both exist, covariance their mean).  One wavefront per pair.
__global__ void __launch_bounds__(64)
k_merge_directions_pnp(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
                       const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
                       const uint32_t* __restrict__ corr, const CorrHeader* __restrict__ hdr,
                       const uint8_t* __restrict__ guided_flag, PassState* __restrict__ fwd,
                       const PassState* __restrict__ back, const uint8_t* __restrict__ mask_f,
                       const uint8_t* __restrict__ mask_b, int min_inliers, int extra_3dof) {
  if ((int)blockIdx.x >= *counter) return;
  const int pair = list[blockIdx.x], lane = threadIdx.x, kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  const CorrHeader h = hdr[pair];
  const bool guided = guided_flag != nullptr && guided_flag[pair] != 0;
  const bool g0 = sf_pnp_dir_gate(0, h, st.meta[sF].x, guided, min_inliers);
  const bool g1 = sf_pnp_dir_gate(1, h, st.meta[sF].x, guided, min_inliers);
  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  const uint32_t* cl = corr + (size_t)pair * kcap;
  int uni = 0, uni_m = 0;
  for (int i = lane; i < kcap; i += 64) uni += (mask_f[(size_t)pair * kcap + i] | mask_b[(size_t)pair * kcap + i]) ? 1 : 0;
  for (int i = lane; i < h.n_corr; i += 64) {
    const uint32_t c = cl[i];
    bool m = false;
    if (g0) { const float* a = xF + 3 * (c & 0xFFFFu); m = sfd::finite3(a[0], a[1], a[2]); }
    if (g1 && !m) { const float* b = xT + 3 * (c >> 16); m = sfd::finite3(b[0], b[1], b[2]); }
    uni_m += m ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { uni += __shfl_xor(uni, off); uni_m += __shfl_xor(uni_m, off); }
  if (lane != 0) return;
  PassState a = fwd[pair];
  const PassState b = back[pair];
  a.inliers = uni;
  a.matches = uni_m;
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
  if (!a.is_null)
    for (int t = 0; t < extra_3dof; ++t) sfd::to3dof_canon(a.T);
  fwd[pair] = a;
}

}  // namespace

size_t sf_pnp_lds_bytes(int kcap, int iterations) { return sf_pnp_lds_bytes_dev(kcap, iterations); }

int sf_launch_pnp(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass) {
  if (n <= 0) return SF_OK;
  const bool ba = c->dparams.bundle_adjustment != 0;
  // (the adjustment is a launch of its own behind the estimate: sf_launch_ba_pass; both directions WITH the adjustment:
  //  plain estimates, then k_merge_directions_ba adjusts over the union)
  const size_t lds = (sf_pnp_lds_bytes(st.kcap, c->dparams.iterations) + 15) & ~(size_t)15;
  if (lds > 160 * 1024) return sf_fail(c, SF_ERANGE, "PnP workgroup needs %zu B of LDS (> 160 KiB)", lds);
  const bool bidir = c->dparams.bidirectional != 0;
  bool& attr = c->pnp_attr_set;
  if (!attr) {   // per handle = per device
    SF_HIP(c, hipFuncSetAttribute((const void*)k_pnp<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SF_HIP(c, hipFuncSetAttribute((const void*)k_pnp<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  const int32_t* list = (const int32_t*)(pass == 1 ? c->list1.p : c->list3.p);
  const int32_t* counter = counters + (pass == 1 ? 0 : 2);
  const uint32_t* corr = (const uint32_t*)(pass == 1 ? c->corr1.p : c->corr2.p);
  const CorrHeader* hdr = (const CorrHeader*)(pass == 1 ? c->hdr1.p : c->hdr2.p);
  PassState* ps = (PassState*)(pass == 1 ? c->pass1.p : c->pass2.p);
  const int end_3dof = c->dparams.force_3dof ? (pass == 1 ? 2 : 1) : 0;
  // (pass 1 always matches globally; pass 2's correspondences are the guided matcher's where its flag says so)
  const uint8_t* guided_flag = pass == 2 ? (const uint8_t*)c->flags.p : nullptr;
  uint8_t *mask_f = nullptr, *mask_b = nullptr;
  if (bidir) {
    int rc;
    const size_t mb = (size_t)n * st.kcap;
    if ((rc = sf_buf_reserve(c, c->dir_mask, 2 * mb)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->pass_back, (size_t)n * sizeof(PassState))) != SF_OK) return rc;
    mask_f = (uint8_t*)c->dir_mask.p;
    mask_b = mask_f + mb;
    SF_HIP(c, hipMemsetAsync(mask_f, 0, 2 * mb, c->stream));
  } else if (ba) {      // the estimate's inlier set, one byte per "from" feature: what the adjustment's launch rebuilds its words from
    int rc;
    const size_t mb = (size_t)n * st.kcap;
    if ((rc = sf_buf_reserve(c, c->dir_mask, mb)) != SF_OK) return rc;
    mask_f = (uint8_t*)c->dir_mask.p;
    SF_HIP(c, hipMemsetAsync(mask_f, 0, mb, c->stream));
  }
  const int kid = pass == 1 ? SF_K_RANSAC1 : SF_K_RANSAC2;
  sf_prof_begin(c, kid);
  auto launch = [&](auto kern, PassState* out, uint8_t* mask, int extra) {
    hipLaunchKernelGGL(kern, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to, list, counter, corr, hdr, out, extra,
                       c->dparams, mask, guided_flag);
  };
  if (ba && !bidir) launch(k_pnp<0>, ps, mask_f, 0);      // (Reg/Force3DoF's end-of-pass application: behind the adjustment)
  else if (!bidir) launch(k_pnp<0>, ps, nullptr, end_3dof);
  else {
    launch(k_pnp<0>, ps, mask_f, 0);
    launch(k_pnp<1>, (PassState*)c->pass_back.p, mask_b, 0);
    if (ba) {
      const int rc = sf_launch_merge_directions_ba(c, st, d_from, d_to, n, pass, true, mask_f, mask_b);
      if (rc != SF_OK) return rc;
    } else
    hipLaunchKernelGGL(k_merge_directions_pnp, dim3(n), dim3(64), 0, c->stream, st, d_from, d_to, list, counter, corr, hdr,
                       guided_flag, ps, (const PassState*)c->pass_back.p, (const uint8_t*)mask_f, (const uint8_t*)mask_b,
                       c->dparams.min_inliers, end_3dof);
  }
  sf_prof_end(c, kid);
  SF_HIP(c, hipGetLastError());
  if (ba && !bidir) return sf_launch_ba_pass(c, st, d_from, d_to, n, pass, pass == 1 ? 1 : 3, mask_f, nullptr, false, nullptr);
  return SF_OK;
}
