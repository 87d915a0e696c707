// k_ransac.hip -- RANSAC 3D->3D rigid registration + model refinement for one candidate pair per
// 256-thread workgroup.
//
// Replaces util3d::estimateMotion3DTo3D as called at myRegistrationVis.cpp:1122-1131 of the
// reference [upstream rtabmap util3d_motion_estimation.cpp / util3d_registration.cpp
// transformFromXYZCorrespondences; PCL RandomSampleConsensus + SampleConsensusModelRegistration +
// the refineModel loop], for the correspondences produced by k_match / k_guided.
//
// CDNA4 mapping:
//   * the finite, non-zero correspondences are gathered once into LDS as float4 pairs;
//   * hypotheses in rounds of 64, one LANE each: wavefront 0 does the stateless keyed sampling, PCL's
//     sample-distance test and the closed-form 3-point rigid fit in fp64 (Horn quaternion) and
//     parks the 64 models in LDS; then all four wavefronts count inliers, each over a quarter of the
//     points, with LDS BROADCAST reads (every lane reads the same address -> one LDS cycle, no bank
//     conflicts) -- 12 fma + compare per point.  With the usual inlier ratios PCL's adaptive rule
//     stops inside the first round, so a round of 64 (not 256) quarters the wasted fits and the
//     point split quarters the counting time;
//   * PCL's sequential adaptive-termination rule is applied afterwards to the per-hypothesis
//     counts, which reproduces the sequential algorithm's choice exactly;
//   * refinement: block-wide fp64 moment reductions in a FIXED order (strided partials, xor
//     butterfly inside each wavefront via DPP shuffles, four wave sums folded left to right), so
//     the result is reproducible bit for bit; inlier sets live in LDS byte masks.
// Compiled with -ffp-contract=off (canonical arithmetic, see sf_device_math.hpp).
#include "sf_device_math.hpp"
#include "sf_internal.hpp"

namespace {

struct RansacLds {
  float4* src;      // [kcap] "from" points (PCL model input_)
  float4* dst;      // [kcap] "to" points   (PCL target_)
  float* d2;        // [kcap] squared residuals of the last selectWithinDistance
  uint8_t* mask_a;  // [kcap]
  uint8_t* mask_b;  // [kcap]
  int* counts;      // [iterations + 2]
  double* red;      // [4][16]
  int* misc;        // [16]
  float* hyp;       // [12][64] models of the current round of hypotheses
  int* hyp_cnt;     // [4][64] partial inlier counts (one row per wavefront) + [64] valid flags
  float* best;      // [12] model of the best hypothesis so far (copied out of `hyp` by the scan)
};

template <int N>
__device__ __forceinline__ void block_sum_vec(double (&v)[N], double* red, int tid) {
  sfd::block_sum_canon<N, 16>(v, red, tid);
}

template <int NW>
__device__ __forceinline__ int block_sum_int(int v, int* misc, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if constexpr (NW == 1) return v;   // one wavefront: the barrier above only orders its LDS traffic
  if (lane == 0) misc[8 + wave] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += misc[8 + w];
  return t;
}

// PCL isSampleGood on the source cloud + keyed sampler; returns false when max_checks attempts fail
__device__ __forceinline__ bool draw_sample(const RansacLds& L, uint64_t seed, uint32_t it, int max_checks,
                                            uint32_t m, double sdt, uint32_t& s0, uint32_t& s1, uint32_t& s2) {
  for (int a = 0; a < max_checks; ++a) {
    sfd::sample_triplet(seed, it, (uint32_t)a, m, s0, s1, s2);
    float4 p0 = L.src[s0], p1 = L.src[s1], p2 = L.src[s2];
    float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
    float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
    float cx = p2.x - p1.x, cy = p2.y - p1.y, cz = p2.z - p1.z;
    float da = (ax * ax + ay * ay) + az * az;
    float db = (bx * bx + by * by) + bz * bz;
    float dc = (cx * cx + cy * cy) + cz * cz;
    if ((double)da > sdt && (double)db > sdt && (double)dc > sdt) return true;
  }
  return false;
}

// 3-point rigid fit, sequential summation order
__device__ inline void fit3(const RansacLds& L, uint32_t s0, uint32_t s1, uint32_t s2, float (&coef)[12]) {
  const float4 p[3] = {L.src[s0], L.src[s1], L.src[s2]};
  const float4 q[3] = {L.dst[s0], L.dst[s1], L.dst[s2]};
  const double inv_n = 1.0 / 3.0;
  double mp[3] = {0.0, 0.0, 0.0}, mq[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mp[0] += (double)p[i].x; mp[1] += (double)p[i].y; mp[2] += (double)p[i].z;
    mq[0] += (double)q[i].x; mq[1] += (double)q[i].y; mq[2] += (double)q[i].z;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { mp[j] *= inv_n; mq[j] *= inv_n; }
  double S[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  double ga = 0.0, gb = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a[3] = {(double)p[i].x - mp[0], (double)p[i].y - mp[1], (double)p[i].z - mp[2]};
    const double b[3] = {(double)q[i].x - mq[0], (double)q[i].y - mq[1], (double)q[i].z - mq[2]};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) S[j][k] += a[j] * b[k];
    ga += (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    gb += (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
  }
  sfd::rigid_from_moments(S, mp, mq, ga, gb, coef);
}

// optimizeModelCoefficients over the members of `mask` (block-order reductions)
template <int NW>
__device__ inline void fit_masked(const RansacLds& L, int m, const uint8_t* mask, int n_in, float (&coef)[12],
                                  int tid) {
  const double inv_n = 1.0 / (double)n_in;
  double s6[6];
  sfd::canon_reduce<6, 16, NW>(m, tid, L.red, s6, [&](int i, double (&a)[6]) {
    if (mask[i]) {
      float4 p = L.src[i], q = L.dst[i];
      a[0] += (double)p.x; a[1] += (double)p.y; a[2] += (double)p.z;
      a[3] += (double)q.x; a[4] += (double)q.y; a[5] += (double)q.z;
    }
  });
  double mp[3] = {s6[0] * inv_n, s6[1] * inv_n, s6[2] * inv_n};
  double mq[3] = {s6[3] * inv_n, s6[4] * inv_n, s6[5] * inv_n};
  double s9[11];   // S (9), ga, gb
  sfd::canon_reduce<11, 16, NW>(m, tid, L.red, s9, [&](int i, double (&acc)[11]) {
    if (mask[i]) {
      float4 p = L.src[i], q = L.dst[i];
      const double a[3] = {(double)p.x - mp[0], (double)p.y - mp[1], (double)p.z - mp[2]};
      const double b[3] = {(double)q.x - mq[0], (double)q.y - mq[1], (double)q.z - mq[2]};
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[3 * j + k] += a[j] * b[k];
      acc[9] += (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
      acc[10] += (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    }
  });
  double S[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) S[j][k] = s9[3 * j + k];
  sfd::rigid_from_moments(S, mp, mq, s9[9], s9[10], coef);
}

// selectWithinDistance: membership mask, member count, and the members' squared residuals in
// L.d2 (non-members and the padding up to a multiple of 4 hold +inf, so order statistics over the
// selected set can scan the array without consulting the mask).
template <int NW>
__device__ inline int select_within(const RansacLds& L, int m, const float (&coef)[12], double thr2,
                                    uint8_t* mask, int tid) {
  int n = 0;
  const int m4 = (m + 3) & ~3;
  for (int i = tid; i < m4; i += 64 * NW) {
    bool in = false;
    float r2 = __int_as_float(0x7F800000);
    if (i < m) {
      float4 p = L.src[i], q = L.dst[i];
      const float v = sfd::residual2(coef, p.x, p.y, p.z, q.x, q.y, q.z);
      in = (double)v < thr2;
      r2 = in ? v : r2;
      mask[i] = in ? 1 : 0;
    }
    L.d2[i] = r2;
    n += in ? 1 : 0;
  }
  return block_sum_int<NW>(n, L.misc, tid);
}

// 2.1981 * median (element n>>1 in sorted order) of the members' squared residuals
// [upstream pcl::SampleConsensusModel::computeVariance].  Exact order statistic by rank counting
// over the +inf-padded residual array (4 values per LDS read, broadcast).
template <int NW>
__device__ inline double variance_of(const RansacLds& L, int m, int n, int tid) {
  const int med = n >> 1;
  const int m4 = (m + 3) & ~3;
  __syncthreads();
  for (int i = tid; i < m; i += 64 * NW) {
    const float v = L.d2[i];
    if (v < __int_as_float(0x7F800000)) {   // member
      int lt = 0, eq = 0;
      const float4* d4 = reinterpret_cast<const float4*>(L.d2);
      for (int j = 0; j < m4 / 4; ++j) {
        const float4 u = d4[j];
        lt += (u.x < v ? 1 : 0) + (u.y < v ? 1 : 0) + (u.z < v ? 1 : 0) + (u.w < v ? 1 : 0);
        eq += (u.x == v ? 1 : 0) + (u.y == v ? 1 : 0) + (u.z == v ? 1 : 0) + (u.w == v ? 1 : 0);
      }
      if (lt <= med && med < lt + eq) L.red[15] = (double)v;  // every writer holds the same value
    }
  }
  __syncthreads();
  const double medv = L.red[15];
  return 2.1981 * medv;
}

// Body of one RANSAC pass for ONE pair (the calling workgroup); smem_raw is the workgroup's dynamic LDS.
// NW = 4: the whole 256-thread workgroup; NW = 1: ONE wavefront runs the pass alone (the other three of the
// workgroup have ended, k_verify.hip) -- same canonical arithmetic (sfd::canon_reduce), same integers.
template <int NW = 4>
__device__ __forceinline__ void ransac_body(const StoreView& st, int pair, const int32_t* __restrict__ pair_from,
                                            const int32_t* __restrict__ pair_to, const uint32_t* __restrict__ corr,
                                            const CorrHeader* __restrict__ hdr, PassState* __restrict__ pass,
                                            const DeviceParams& P, unsigned char* smem_raw, int trace_base = 2) {
  constexpr int NT = 64 * NW;
  // NW < 4: the live wavefronts are (pair & 3), (pair & 3) + 1, ... mod 4 (k_verify_fused), numbered from 0 here
  const int tid = NW == 4 ? (int)threadIdx.x
                          : (int)((((threadIdx.x >> 6) - (unsigned)(pair & 3)) & 3u) * 64u + (threadIdx.x & 63u));
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];

  RansacLds L;
  {
    unsigned char* p = smem_raw;
    L.src = (float4*)p; p += (size_t)kcap * 16;
    L.dst = (float4*)p; p += (size_t)kcap * 16;
    L.red = (double*)p; p += 64 * 8;
    L.d2 = (float*)p; p += (size_t)kcap * 4;
    L.counts = (int*)p; p += (size_t)((P.iterations + 2 + 3) & ~3) * 4;
    L.misc = (int*)p; p += 16 * 4;
    L.mask_a = p; p += kcap;
    L.mask_b = p; p += kcap;
    L.hyp = (float*)p; p += 12 * 64 * 4;
    L.hyp_cnt = (int*)p; p += 5 * 64 * 4;
    L.best = (float*)p;
  }

  // ---- util3d::findCorrespondences: finite, non-zero, id-ordered ---------------------------------
  const int n_corr = hdr[pair].n_corr;
  const uint32_t* cl = corr + (size_t)pair * kcap;
  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float* xT = st.xyz + (size_t)sT * kcap * 3;
  if (tid < 16) L.misc[tid] = 0;
  __syncthreads();
  int m = 0;
  for (int base = 0; base < n_corr; base += NT) {
    const int i = base + tid;
    bool ok = false;
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
    if (i < n_corr) {
      const uint32_t c = cl[i];
      const float* a = xF + 3 * (c & 0xFFFFu);
      const float* b = xT + 3 * (c >> 16);
      ax = a[0]; ay = a[1]; az = a[2];
      bx = b[0]; by = b[1]; bz = b[2];
      ok = sfd::finite3(ax, ay, az) && sfd::finite3(bx, by, bz) && (ax != 0.f || ay != 0.f || az != 0.f) &&
           (bx != 0.f || by != 0.f || bz != 0.f);
    }
    const unsigned long long bal = __ballot(ok);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) L.misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int c = L.misc[4 + w];
      if (w < wave) woff += c;
      total += c;
    }
    if (ok) {
      L.src[m + woff + before] = make_float4(ax, ay, az, 0.f);
      L.dst[m + woff + before] = make_float4(bx, by, bz, 0.f);
    }
    m += total;
    __syncthreads();
  }

  SF_TRACE_MARK(P, pair, trace_base + 0);
  PassState ps;
#pragma unroll
  for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
  ps.var = 1.0; ps.var_ang = 1.0;
  ps.is_null = 1;
  ps.inliers = 0;
  ps.matches = m;
  ps.pad = 0;
  if (m < P.min_inliers || m < 3) {
    if (tid == 0) pass[pair] = ps;
    return;
  }

  // ---- computeSampleDistanceThreshold (PCA of the source cloud) ----------------------------------
  const double inv_m = 1.0 / (double)m;
  double mean[3];
  {
    double s3[3];
    sfd::canon_reduce<3, 16, NW>(m, tid, L.red, s3, [&](int i, double (&a)[3]) {
      float4 p = L.src[i];
      a[0] += (double)p.x; a[1] += (double)p.y; a[2] += (double)p.z;
    });
    mean[0] = s3[0] * inv_m; mean[1] = s3[1] * inv_m; mean[2] = s3[2] * inv_m;
  }
  double sdt;
  {
    double c6[6];  // xx xy xz yy yz zz
    sfd::canon_reduce<6, 16, NW>(m, tid, L.red, c6, [&](int i, double (&a)[6]) {
      float4 p = L.src[i];
      const double a0 = (double)p.x - mean[0], a1 = (double)p.y - mean[1], a2 = (double)p.z - mean[2];
      a[0] += a0 * a0; a[1] += a0 * a1; a[2] += a0 * a2;
      a[3] += a1 * a1; a[4] += a1 * a2; a[5] += a2 * a2;
    });
    double ev[3];
    sfd::sym3_eigenvalues(c6[0] * inv_m, c6[1] * inv_m, c6[2] * inv_m, c6[3] * inv_m, c6[4] * inv_m, c6[5] * inv_m, ev);
    sdt = ((sqrt(ev[0]) + sqrt(ev[1])) + sqrt(ev[2])) / 3.0;
    sdt = sdt * sdt;
  }

  if (P.dbg_stop == 1) { if (tid == 0) pass[pair] = ps; return; }   // diagnostic truncation (SF_RANSAC_STOP)
  SF_TRACE_MARK(P, pair, trace_base + 1);

  // ---- hypotheses: one lane each -----------------------------------------------------------------
  const double thr = P.inlier_thr;
  const double thr2 = thr * thr;
  float thr2f = (float)thr2;                       // largest float strictly below thr2
  if ((double)thr2f >= thr2) thr2f = __uint_as_float(__float_as_uint(thr2f) - 1u);
  const int max_it = P.iterations;
  // Hypotheses are evaluated in rounds of 64 (one lane each); after every round thread 0 advances
  // PCL's sequential loop (adaptive k) over the counts available so far.  Once that loop has
  // terminated the remaining rounds are skipped: their counts would never be read.
  double k_adapt = 1.0;                                  // thread 0 only
  const double log_probability = sfd::canon_log(1.0 - 0.99);
  int sc_best = -1, sc_best_it = -1, sc_it = 0;          // thread 0 only
  const int slice = (m + NW - 1) / NW;                   // points counted by each wavefront
  const int i0 = min(m, wave * slice), i1 = min(m, i0 + slice);
  int* hv = L.hyp_cnt + 4 * 64;
  // the wavefront that samples and fits rotates with the pair, so that the workgroups sharing a CU do
  // not all put this serial fp64 section on the same SIMD
  const int fit_wave = NW == 4 ? (pair & 3) : 0;   // (NW < 4: the live wavefronts already rotate with the pair)
  for (int base = 0; base <= max_it; base += 64) {
    const int it = base + lane;
    if (wave == fit_wave) {
      int valid = 0;
      if (it <= max_it) {
        uint32_t s0, s1, s2;
        if (draw_sample(L, P.seed, (uint32_t)it, P.max_sample_checks, (uint32_t)m, sdt, s0, s1, s2)) {
          float coef[12];
          fit3(L, s0, s1, s2, coef);
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
        for (int i = i0; i < i1; ++i) {
          const float4 p = L.src[i], q = L.dst[i];  // same address in every lane: LDS broadcast
          const float r2 = sfd::residual2(coef, p.x, p.y, p.z, q.x, q.y, q.z);
          cnt += (r2 <= thr2f) ? 1 : 0;
        }
      }
      L.hyp_cnt[wave * 64 + lane] = cnt;
    }
    __syncthreads();
    if (wave == fit_wave && it <= max_it)
    {
      int tot = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += L.hyp_cnt[w * 64 + lane];
      L.counts[it] = hv[lane] ? tot : -1;
    }
    __syncthreads();
    if (tid == 0) {
      const int lim = min(max_it, base + 63);  // last iteration whose count exists
      bool stop = false;
      while (true) {
        if (P.adaptive_stop && !((double)sc_it < k_adapt)) { stop = true; break; }
        if (sc_it > max_it) { stop = true; break; }
        if (sc_it > lim) break;                           // needs the next round
        const int c = L.counts[sc_it];
        if (c < 0) { stop = true; break; }                // getSamples failed -> PCL breaks out
        if (c > sc_best) {
          sc_best = c;
          sc_best_it = sc_it;
#pragma unroll
          for (int k = 0; k < 12; ++k) L.best[k] = L.hyp[k * 64 + (sc_it - base)];   // keep the winning model
          const double w = (double)sc_best * inv_m;
          double pno = 1.0 - (w * w) * w;
          if (pno < 2.220446049250313e-16) pno = 2.220446049250313e-16;
          if (pno > 1.0 - 2.220446049250313e-16) pno = 1.0 - 2.220446049250313e-16;
          k_adapt = log_probability / sfd::canon_log(pno);
        }
        ++sc_it;
        if (sc_it > max_it) { stop = true; break; }
      }
      L.misc[0] = sc_best_it;
      L.misc[1] = stop ? 1 : 0;
    }
    __syncthreads();
    if (L.misc[1]) break;
  }
  if (P.dbg_stop == 2 || P.dbg_stop == 3) { if (tid == 0) pass[pair] = ps; return; }   // diagnostic truncation
  SF_TRACE_MARK(P, pair, trace_base + 2);
  const int best_it = L.misc[0];
  if (best_it < 0) {
    if (tid == 0) pass[pair] = ps;
    return;
  }

  if (P.dbg_stop == 3) { if (tid == 0) pass[pair] = ps; return; }   // diagnostic truncation (SF_RANSAC_STOP)

  // ---- winning model (parked in LDS by the scan) --------------------------------------------------
  float coef[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) coef[k] = L.best[k];
  uint8_t* inl = L.mask_a;
  int n_inl = select_within<NW>(L, m, coef, thr2, inl, tid);
  int n_last = n_inl;

  if (P.dbg_stop == 4) { if (tid == 0) pass[pair] = ps; return; }   // diagnostic truncation (SF_RANSAC_STOP)
  SF_TRACE_MARK(P, pair, trace_base + 3);

  // ---- refine loop (copy of pcl::SampleConsensus::refineModel inside rtabmap) ----------------------
  if (P.refine_iterations > 0) {
    double error_threshold = thr;
    int refine_iterations = 0;
    bool inlier_changed = false;
    uint8_t* prev = L.mask_a;
    uint8_t* neu = L.mask_b;
    int n_prev = n_inl, n_new = 0;
    for (int i = tid; i < m; i += NT) neu[i] = 0;
    int n_sizes = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0;  // last four pushed sizes (z1 newest)
    float newc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) newc[i] = coef[i];
    do {
      if (n_prev >= 3) fit_masked<NW>(L, m, prev, n_prev, newc, tid);
      z4 = z3; z3 = z2; z2 = z1; z1 = n_prev;
      ++n_sizes;
      __syncthreads();
      n_new = select_within<NW>(L, m, newc, error_threshold * error_threshold, neu, tid);
      n_last = n_new;
      if (n_new == 0) {
        ++refine_iterations;
        if (refine_iterations >= P.refine_iterations) break;
        continue;
      }
      const double variance = variance_of<NW>(L, m, n_new, tid);
      const double sthr = P.refine_sigma * sqrt(variance);
      error_threshold = thr < sthr ? thr : sthr;
      inlier_changed = false;
      { uint8_t* t = prev; prev = neu; neu = t; int tn = n_prev; n_prev = n_new; n_new = tn; }
      if (n_new != n_prev) {
        if (n_sizes >= 4 && z1 == z3 && z2 == z4) break;  // oscillating
        inlier_changed = true;
        continue;
      }
      int diff = 0;
      for (int i = tid; i < m; i += NT) diff |= (prev[i] != neu[i]) ? 1 : 0;
      inlier_changed = block_sum_int<NW>(diff, L.misc, tid) != 0;
    } while (inlier_changed && ++refine_iterations < P.refine_iterations);
    inl = neu;
    n_inl = n_new;
#pragma unroll
    for (int i = 0; i < 12; ++i) coef[i] = newc[i];
  }

  if (P.dbg_stop == 5) { if (tid == 0) pass[pair] = ps; return; }
  SF_TRACE_MARK(P, pair, trace_base + 4);
  if (n_inl >= 3) {
    const double variance = variance_of<NW>(L, m, n_last, tid);
    ps.var = variance;
    ps.var_ang = variance;
    ps.inliers = n_inl;
    if (n_inl >= P.min_inliers) {
      double R[9], t[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (double)coef[4 * i + j];
        t[i] = (double)coef[4 * i + 3];
      }
      bool allz = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) ps.T[4 * i + j] = (float)R[3 * j + i];
        ps.T[4 * i + 3] = (float)(-((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]));
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) allz = allz && (ps.T[i] == 0.f);
      ps.is_null = allz ? 1 : 0;
    }
  }
  if (tid == 0) pass[pair] = ps;
  SF_TRACE_MARK(P, pair, trace_base + 5);
}

__global__ void __launch_bounds__(SF_BLOCK, 4)
k_ransac(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
         const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
         const uint32_t* __restrict__ corr, const CorrHeader* __restrict__ hdr,
         PassState* __restrict__ pass, DeviceParams P) {
  if ((int)blockIdx.x >= *counter) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ransac_body<4>(st, list[blockIdx.x], pair_from, pair_to, corr, hdr, pass, P, smem_raw);
}

}  // namespace

size_t sf_ransac_lds_bytes(int kcap, int iterations) {
  return (size_t)kcap * 32 + 64 * 8 + (size_t)kcap * 4 + (size_t)((iterations + 2 + 3) & ~3) * 4 + 16 * 4 +
         (size_t)kcap * 2 + 12 * 64 * 4 + 5 * 64 * 4 + 16 * 4;
}

int sf_launch_ransac(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass) {
  if (n <= 0) return SF_OK;
  const size_t lds = sf_ransac_lds_bytes(st.kcap, c->dparams.iterations);
  if (lds > 160 * 1024) return sf_fail(c, SF_ERANGE, "RANSAC workgroup needs %zu B of LDS (> 160 KiB)", lds);
  if (!c->ransac_attr_set) {   // per handle = per device
    SF_HIP(c, hipFuncSetAttribute((const void*)k_ransac, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->ransac_attr_set = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  const int kid = pass == 1 ? SF_K_RANSAC1 : SF_K_RANSAC2;
  sf_prof_begin(c, kid);
  hipLaunchKernelGGL(k_ransac, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                     (const int32_t*)(pass == 1 ? c->list1.p : c->list3.p), counters + (pass == 1 ? 0 : 2),
                     (const uint32_t*)(pass == 1 ? c->corr1.p : c->corr2.p),
                     (const CorrHeader*)(pass == 1 ? c->hdr1.p : c->hdr2.p),
                     (PassState*)(pass == 1 ? c->pass1.p : c->pass2.p), c->dparams);
  sf_prof_end(c, kid);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
