// k_ransac.hip -- RANSAC 3D->3D rigid registration + model refinement for one candidate pair per
// 256-thread workgroup.
//
// Replaces util3d::estimateMotion3DTo3D as called at myRegistrationVis.cpp:1122-1131 of the
// reference [upstream rtabmap util3d_motion_estimation.cpp / util3d_registration.cpp
// transformFromXYZCorrespondences; PCL RandomSampleConsensus + SampleConsensusModelRegistration +
// the refineModel loop], for the correspondences produced by k_match / k_guided.
//
// CDNA4 mapping (round 2: the chain of a surviving pair is the latency tail of the fused kernel, so every phase
// is laid out for few dependent steps and few issued instructions -- results are unchanged, bit for bit):
//   * the finite, non-zero correspondences are gathered once into LDS as float4 pairs;
//   * hypotheses in rounds, one LANE each: ONE wavefront samples (stateless keyed sampler, PCL's sample-distance
//     test) and fits (closed-form 3-point rigid fit in fp64, Horn quaternion) and parks the models in LDS; then
//     all four wavefronts count inliers, each over a quarter of the points.  The first round has 16 hypotheses
//     (PCL's adaptive rule stops after a handful of iterations when the correspondences are good), counted by
//     four lanes per hypothesis; later rounds have 64.  Point loads are issued four at a time so the counting
//     loop is not one LDS round trip per point;
//   * PCL's sequential adaptive-termination rule is replayed over a round's counts by a wavefront SCAN (prefix
//     maximum = the best count after every iteration, the bound k after every iteration evaluated by all lanes at
//     once, first lane whose top-of-loop test fails = where the sequential loop stops): the same decisions as the
//     sequential replay, one logarithm deep instead of one per improvement;
//   * refinement: block-wide fp64 moment reductions in a FIXED order (sfd::canon_reduce); the rigid fit of the
//     reduced moments is solved by ONE wavefront and broadcast through LDS (the other three wait at the barrier
//     instead of issuing the same ~500 fp64 instructions); selection, membership change and the 3-sigma threshold
//     test share one pass and one block reduction; the exact median of the residuals (rank counting) is only
//     evaluated when the threshold actually shrinks, and once for the covariance.
// Compiled with -ffp-contract=off (canonical arithmetic, see sf_device_math.hpp).
#include "sf_device_math.hpp"
#include "sf_internal.hpp"
#include "k_ba.hip"

namespace {

struct RansacLds {
  float4* src;      // [kcap] "from" points (PCL model input_)
  float4* dst;      // [kcap] "to" points   (PCL target_)
  float* d2;        // [kcap] squared residuals of the last selectWithinDistance
  uint8_t* mask_a;  // [kcap]
  uint8_t* mask_b;  // [kcap]
  double* red;      // [4][16]
  unsigned long long* sums;   // [4] packed per-wavefront counters of a selection
  int* misc;        // [16]
  float* hyp;       // [12][64] models of the current round of hypotheses
  int* hyp_cnt;     // [4][64] inlier counts (first round: one row of partial counts per wavefront; later rounds: the
                    // counts of the 64 hypotheses each wavefront fitted itself, -1 = no sample) + [64] valid flags of the first round
  float* best;      // [12] model of the best hypothesis so far (copied out of `hyp` by the scan)
  float* bc;        // [12] model being refined (broadcast from the solving wavefront)
  uint32_t* cidx;   // [kcap] packed (from | to << 16) feature indices of the gathered correspondences (bundle adjustment)
};

__device__ __forceinline__ RansacLds ransac_carve(unsigned char* p, int kcap) {
  RansacLds L;
  L.src = (float4*)p; p += (size_t)kcap * 16;
  L.dst = (float4*)p; p += (size_t)kcap * 16;
  L.red = (double*)p; p += 64 * 8;
  L.sums = (unsigned long long*)p; p += 4 * 8;
  L.d2 = (float*)p; p += (size_t)kcap * 4;
  L.misc = (int*)p; p += 16 * 4;
  L.mask_a = p; p += kcap;
  L.mask_b = p; p += kcap;
  L.hyp = (float*)p; p += 12 * 64 * 4;
  L.hyp_cnt = (int*)p; p += 5 * 64 * 4;
  L.best = (float*)p; p += 16 * 4;
  L.bc = (float*)p; p += 16 * 4;
  L.cidx = (uint32_t*)p;
  return L;
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

// optimizeModelCoefficients over the members of `mask` (block-order reductions); the fit of the reduced moments is
// solved by wavefront 0 and left in L.bc (visible to the workgroup after the caller's next barrier)
template <int NW>
__device__ inline void fit_masked(const RansacLds& L, int m, const uint8_t* mask, int n_in, int tid) {
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
  if (tid < 64) {
    double S[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) S[j][k] = s9[3 * j + k];
    float coef[12];
    sfd::rigid_from_moments(S, mp, mq, s9[9], s9[10], coef);
    if (tid < 12) {
      float v = coef[0];
#pragma unroll
      for (int k = 1; k < 12; ++k) v = (tid == k) ? coef[k] : v;
      L.bc[tid] = v;
    }
  }
}

// Result of one selectWithinDistance pass (reduced together, 13 bits per field, K <= 4096):
//   n    members (squared residual < thr2)
//   low  members whose residual keeps the 3-sigma rule BELOW the inlier threshold, i.e. for which
//        !(thr < sigma * sqrt(2.1981 * d2)) -- the rule's operations are monotone in d2, so these are the `low`
//        smallest members and the median (rank n >> 1) is one of them iff n >> 1 < low
//   diff elements whose membership differs from `prev`
struct SelCounts { int n, low, diff; };

// selectWithinDistance under the model at `cf` (LDS): membership mask, the members' squared residuals in L.d2
// (non-members and the padding up to a multiple of 4 hold +inf, so order statistics over the selected set can
// scan the array without consulting the mask), and the three counts above in ONE block reduction.
template <int NW>
__device__ inline SelCounts select_within(const RansacLds& L, int m, const float* cf, double thr2, uint8_t* mask,
                                          const uint8_t* prev, double sigma, double thr, int tid) {
  float coef[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) coef[k] = cf[k];
  unsigned long long acc = 0;
  const int m4 = (m + 3) & ~3;
  for (int i = tid; i < m4; i += 64 * NW) {
    float r2 = __int_as_float(0x7F800000);
    if (i < m) {
      float4 p = L.src[i], q = L.dst[i];
      const float v = sfd::residual2(coef, p.x, p.y, p.z, q.x, q.y, q.z);
      const bool in = (double)v < thr2;
      r2 = in ? v : r2;
      const bool was = prev ? prev[i] != 0 : false;
      mask[i] = in ? 1 : 0;
      const bool low = in && !(thr < sigma * sqrt(2.1981 * (double)v));
      acc += (in ? 1ull : 0ull) + (low ? (1ull << 13) : 0ull) + ((was != in) ? (1ull << 26) : 0ull);
    }
    L.d2[i] = r2;
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  __syncthreads();
  if (lane == 0) L.sums[wave] = acc;
  __syncthreads();
  unsigned long long t = L.sums[0];            // (integer fields: the order of the fold is free)
#pragma unroll
  for (int w = 1; w < NW; ++w) t += L.sums[w];
  SelCounts c;
  c.n = (int)(t & 0x1FFFull);
  c.low = (int)((t >> 13) & 0x1FFFull);
  c.diff = (int)((t >> 26) & 0x1FFFull);
  return c;
}

// 2.1981 * median (element n>>1 in sorted order) of the members' squared residuals
// [upstream pcl::SampleConsensusModel::computeVariance].  Exact order statistic by a most-significant-digit radix
// select over the +inf-padded residual array: the residuals are non-negative floats, whose bit patterns order like
// unsigned integers, non-members hold +inf and sort behind every member, so the member of rank n >> 1 is the
// element of that rank in the whole array.  Four passes of an 8-bit histogram (LDS atomics; the four histograms live
// in the hypothesis models of the finished search, L.hyp .. L.hyp_cnt) and a 256-bin scan by one wavefront: O(m)
// work, ~200 instructions per wavefront -- round 1 counted ranks, O(m^2): ~2 000 instructions per wavefront and call,
// a quarter of the whole chain's vector instructions.
// (a real call, LDS pointers in their own address space: inlined at its two call sites the select tipped the
//  register allocation of the fused kernel into spilling around the hypothesis loop)
typedef __attribute__((address_space(3))) const float* lds_cfloat_p;
typedef __attribute__((address_space(3))) unsigned* lds_uint_p;
typedef __attribute__((address_space(3))) int* lds_int_p;
template <int NT>
__device__ __attribute__((noinline)) float radix_select_rank(lds_cfloat_p d2, lds_uint_p hist, lds_int_p misc, int m4,
                                                             int rank, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int b = tid; b < 4 * 256; b += NT) hist[b] = 0u;
  __syncthreads();                                       // histograms cleared, the selection's d2 visible
  unsigned prefix = 0u;
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const int shift = 24 - 8 * p;
    for (int i = tid; i < m4; i += NT) {
      const unsigned bits = __float_as_uint(d2[i]);
      if (p == 0 || (bits >> (shift + 8)) == prefix)
        __hip_atomic_fetch_add(&hist[p * 256 + ((bits >> shift) & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (wave == 0) {
      const unsigned h0 = hist[p * 256 + 4 * lane], h1 = hist[p * 256 + 4 * lane + 1], h2 = hist[p * 256 + 4 * lane + 2],
                     h3 = hist[p * 256 + 4 * lane + 3];
      const unsigned s = (h0 + h1) + (h2 + h3);
      unsigned incl = s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
      }
      const unsigned excl = incl - s;
      if (excl <= (unsigned)rank && (unsigned)rank < incl) {   // exactly one lane
        unsigned r = (unsigned)rank - excl;
        int b = 0;
        if (r >= h0) { r -= h0; b = 1; if (r >= h1) { r -= h1; b = 2; if (r >= h2) { r -= h2; b = 3; } } }
        misc[2] = (int)((prefix << 8) | (unsigned)(4 * lane + b));
        misc[3] = (int)r;
      }
    }
    __syncthreads();
    prefix = (unsigned)misc[2];
    rank = misc[3];
  }
  return __uint_as_float(prefix);
}

template <int NW>
__device__ __forceinline__ double variance_of(const RansacLds& L, int m, int n, int tid) {
  const float medv = radix_select_rank<64 * NW>((lds_cfloat_p)L.d2, (lds_uint_p) reinterpret_cast<unsigned*>(L.hyp),
                                       (lds_int_p)L.misc, (m + 3) & ~3, n >> 1, tid);
  return 2.1981 * (double)medv;
}

// Inlier counts of a round of R hypotheses (R = 16: four lanes per hypothesis, R = 64: one) over this wavefront's
// quarter [i0, i1) of the points; point loads are issued four at a time (same address in the lanes of a group:
// LDS broadcast).  Returns this lane's partial count.
template <int R>
__device__ __forceinline__ int count_round(const RansacLds& L, const int* hv, int i0, int i1, int lane, float thr2f) {
  constexpr int G = 64 / R;
  const int h = lane & (R - 1), g = lane / R;
  int cnt = 0;
  if (hv[h]) {
    float coef[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) coef[k] = L.hyp[k * 64 + h];
    int i = i0 + g;
#pragma unroll 1
    for (; i + 3 * G < i1; i += 4 * G) {
      float4 p[4], q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { p[u] = L.src[i + u * G]; q[u] = L.dst[i + u * G]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float r2 = sfd::residual2(coef, p[u].x, p[u].y, p[u].z, q[u].x, q[u].y, q[u].z);
        cnt += (r2 <= thr2f) ? 1 : 0;
      }
    }
#pragma unroll 1
    for (; i < i1; i += G) {
      const float4 p = L.src[i], q = L.dst[i];
      const float r2 = sfd::residual2(coef, p.x, p.y, p.z, q.x, q.y, q.z);
      cnt += (r2 <= thr2f) ? 1 : 0;
    }
  }
  return cnt;
}

// Inlier count of THIS lane's hypothesis (coefficients in registers) over all m points: the rounds behind the first,
// where every wavefront fits and counts 64 hypotheses of its own.  Point loads four at a time (LDS broadcast).
__device__ __forceinline__ int count_own(const RansacLds& L, const float (&coef)[12], int m, float thr2f) {
  int cnt = 0;
  int i = 0;
#pragma unroll 1
  for (; i + 1 < m; i += 2) {
    const float4 p0 = L.src[i], q0 = L.dst[i], p1 = L.src[i + 1], q1 = L.dst[i + 1];
    const float r0 = sfd::residual2(coef, p0.x, p0.y, p0.z, q0.x, q0.y, q0.z);
    const float r1 = sfd::residual2(coef, p1.x, p1.y, p1.z, q1.x, q1.y, q1.z);
    cnt += ((r0 <= thr2f) ? 1 : 0) + ((r1 <= thr2f) ? 1 : 0);
  }
  if (i < m) {
    const float4 p = L.src[i], q = L.dst[i];
    const float r2 = sfd::residual2(coef, p.x, p.y, p.z, q.x, q.y, q.z);
    cnt += (r2 <= thr2f) ? 1 : 0;
  }
  return cnt;
}

// State of PCL's sequential loop between rounds (wave-uniform in the solving wavefront)
struct ScanState {
  int best;      // best inlier count so far (-1: none)
  int best_it;   // its iteration
  int it;        // iterations processed (PCL's iterations_)
  double k;      // adaptive bound
};

// __shfl_up with the source lane derived from the CALLER's lane number: the library form derives it from the hardware
// lane id, which is loop invariant -- the six source addresses of the scan below were hoisted in front of the
// hypothesis loop and spilled around the fit.
__device__ __forceinline__ int shfl_up_from(int v, int off, int lane) {
  const int src = lane >= off ? lane - off : lane;
  return __builtin_amdgcn_ds_bpermute(src << 2, v);
}
__device__ __forceinline__ double shfl_up_from(double v, int off, int lane) {
  const int src = lane >= off ? lane - off : lane;
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xFFFFFFFFll));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Advance PCL's loop [upstream pcl::RandomSampleConsensus::computeModel] over the R counts of one round, held one
// per lane (c = count of iteration base + lane, -1 when getSamples failed; `present` = lane < R and the
// iteration is <= max_it).  Sequentially the loop tests at the top of iteration `it`:  (adaptive && !(it < k)) ->
// stop; it > max_it -> stop; count < 0 -> stop; then updates best / k on a STRICT improvement and increments it.
// k is a function of the best count alone, so the value it has after iteration j is f(max(best, c_0 .. c_j)): every
// lane evaluates f on its inclusive prefix maximum, the top-of-loop test of lane j uses lane j-1's value, and the
// first lane whose test fails is where the sequential loop stops.  Returns true when the loop has terminated.
// park: the round's models sit in L.hyp (first round) and the winner's is copied to L.best here; otherwise the lane
// that owns the winning iteration writes it after the replay (ransac_body).
__device__ __forceinline__ bool replay_round(const RansacLds& L, ScanState& S, int base, int R, int lane, int c,
                                             bool present, int max_it, bool adaptive, double inv_m,
                                             double log_probability, bool park = true) {
  int x = present ? c : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = shfl_up_from(x, off, lane);
    if (lane >= off) x = max(x, y);
  }
  x = max(x, S.best);                                   // inclusive prefix maximum incl. the earlier rounds
  double kj = 1.0;                                      // (no best yet: k keeps its initial value)
  if (x >= 0) {
    const double w = (double)x * inv_m;
    double pno = 1.0 - (w * w) * w;
    if (pno < 2.220446049250313e-16) pno = 2.220446049250313e-16;
    if (pno > 1.0 - 2.220446049250313e-16) pno = 1.0 - 2.220446049250313e-16;
    kj = log_probability / sfd::canon_log(pno);
  }
  double kprev = shfl_up_from(kj, 1, lane);
  if (lane == 0) kprev = S.k;
  const int it = base + lane;
  const bool top_ok = present && (!adaptive || (double)it < kprev) && c >= 0;
  const unsigned long long fail = ~__ballot(top_ok);
  const int n_proc = fail ? (__ffsll((long long)fail) - 1) : 64;
  if (n_proc > 0) {
    const int nb = __builtin_amdgcn_readlane(x, n_proc - 1);          // n_proc comes from a ballot: wave uniform
    const long long kb = __double_as_longlong(kj);
    const double nk = __longlong_as_double(
        ((long long)__builtin_amdgcn_readlane((int)(kb >> 32), n_proc - 1) << 32) |
        (unsigned)__builtin_amdgcn_readlane((int)(kb & 0xFFFFFFFFll), n_proc - 1));
    if (nb > S.best) {
      const unsigned long long at = __ballot(present && c == nb && lane < n_proc);
      const int lb = __ffsll((long long)at) - 1;        // first iteration that reached the final best
      S.best = nb;
      S.best_it = base + lb;
      if (park && lane < 12) L.best[lane] = L.hyp[lane * 64 + lb];   // keep the winning model
    }
    S.k = nk;
  }
  S.it = base + n_proc;
  return n_proc < R || base + R > max_it;
}

__host__ __device__ inline size_t sf_ransac_lds_bytes_dev(int kcap) {
  return (size_t)kcap * 32 + 64 * 8 + 4 * 8 + (size_t)kcap * 4 + 16 * 4 + (size_t)kcap * 2 + 12 * 64 * 4 + 5 * 64 * 4 +
         16 * 4 + 16 * 4 + (size_t)kcap * 4;
}

// The "no transform" state of a pass (identity covariance scale, `matches` correspondences seen): built where it is
// written, so that no copy of it stays live in registers across the pass.
__device__ __forceinline__ void write_null_pass(PassState& out, int matches) {
  PassState ps;
#pragma unroll
  for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
  ps.var = 1.0; ps.var_ang = 1.0;
  ps.is_null = 1;
  ps.inliers = 0;
  ps.matches = matches;
  ps.pad = 0;
  out = ps;
}

// Body of one RANSAC pass for ONE pair (the calling 256-thread workgroup).  `cl` = the pair's n_corr
// correspondences (from | to << 16, ascending "from"; LDS in the fused kernel, global in the stage kernel);
// `lds` = this stage's region of the workgroup's dynamic LDS (sf_ransac_lds_bytes).  The result is written to
// `out` by thread 0.
// DIR = 1: the backward estimate of Vis/ForwardEstOnly = false (myRegistrationVis.cpp:936-978: A = "to", B = "from");
// `mask_out` (stage kernel, DIR form only): one byte per "from" feature of the pair, set for this estimate's inliers.
template <int DIR = 0, int NW = 4>
__device__ __forceinline__ void ransac_body(const StoreView& st, int pair, int sF, int sT, const uint32_t* cl, int n_corr,
                                            PassState& out, const DeviceParams& P, unsigned char* lds,
                                            int trace_base = 2, uint8_t* mask_out = nullptr) {
  constexpr int NT = 64 * NW;      // NW = 4: the 256-thread workgroup of the fused / stage kernels; 1, 2: the low-occupancy chains
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const RansacLds L = ransac_carve(lds, kcap);

  // ---- util3d::findCorrespondences: finite, non-zero, id-ordered ---------------------------------
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
      const float* a = DIR ? xT + 3 * (c >> 16) : xF + 3 * (c & 0xFFFFu);
      const float* b = DIR ? xF + 3 * (c & 0xFFFFu) : xT + 3 * (c >> 16);
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
      L.cidx[m + woff + before] = cl[i];
    }
    m += total;
    __syncthreads();
  }

  SF_TRACE_MARK(P, pair, trace_base + 0);
  if (m < P.min_inliers || m < 3) {
    if (tid == 0) write_null_pass(out, m);
    return;
  }

  // the wavefront that samples, fits and replays rotates with the pair, so that the workgroups sharing a CU do
  // not all put this serial fp64 section on the same SIMD
  const int fit_wave = pair & (NW - 1);

  // ---- computeSampleDistanceThreshold (PCA of the source cloud) ----------------------------------
  const double inv_m = 1.0 / (double)m;
  if (tid == 0) reinterpret_cast<double*>(L.sums)[2] = inv_m;   // (the replay reads it back: not worth two registers
                                                                //  across the whole hypothesis loop)
  double sdt = 0.0;     // only the solving wavefront needs (and computes) it; kept in LDS from here on
  {
    double mean[3];
    double s3[3];
    sfd::canon_reduce<3, 16, NW>(m, tid, L.red, s3, [&](int i, double (&a)[3]) {
      float4 p = L.src[i];
      a[0] += (double)p.x; a[1] += (double)p.y; a[2] += (double)p.z;
    });
    mean[0] = s3[0] * inv_m; mean[1] = s3[1] * inv_m; mean[2] = s3[2] * inv_m;
    double c6[6];  // xx xy xz yy yz zz
    sfd::canon_reduce<6, 16, NW>(m, tid, L.red, c6, [&](int i, double (&a)[6]) {
      float4 p = L.src[i];
      const double a0 = (double)p.x - mean[0], a1 = (double)p.y - mean[1], a2 = (double)p.z - mean[2];
      a[0] += a0 * a0; a[1] += a0 * a1; a[2] += a0 * a2;
      a[3] += a1 * a1; a[4] += a1 * a2; a[5] += a2 * a2;
    });
    if (wave == fit_wave) {
      double ev[3];
      sfd::sym3_eigenvalues(c6[0] * inv_m, c6[1] * inv_m, c6[2] * inv_m, c6[3] * inv_m, c6[4] * inv_m, c6[5] * inv_m, ev);
      sdt = ((sqrt(ev[0]) + sqrt(ev[1])) + sqrt(ev[2])) / 3.0;
      sdt = sdt * sdt;
      // the threshold and the state of PCL's loop are parked in LDS between rounds (L.sums is free until the first
      // selection): they are only touched by the solving wavefront, a few times per round, and would otherwise
      // occupy registers across the fit
      if (lane == 0) {
        double* st = reinterpret_cast<double*>(L.sums);
        st[0] = sdt;       // sample-distance threshold
        st[1] = 1.0;       // ScanState.k
        L.misc[12] = -1;   // ScanState.best
        L.misc[13] = -1;   // ScanState.best_it
        L.misc[14] = 0;    // ScanState.it
      }
    }
  }

  if (P.dbg_stop == 1) { if (tid == 0) write_null_pass(out, m); return; }   // diagnostic truncation (SF_RANSAC_STOP)
  SF_TRACE_MARK(P, pair, trace_base + 1);

  // ---- hypotheses: one lane each; a first round of 16, then rounds of 64 per WAVEFRONT -----------------------------
  // (values only needed behind this loop -- thr, thr2, sigma -- are formed there, and the replay's constants inside
  //  the replay: the fit in the middle of the loop wants every register it can get)
  const int max_it = P.iterations;
  int* hv = L.hyp_cnt + 4 * 64;
  {
    // First round: iterations 0..15, fitted by ONE wavefront and counted by all of them, four lanes per hypothesis (PCL's
    // adaptive rule stops after a handful of iterations when the correspondences are good: everything is laid out for
    // that round to be short).
    // The lane / wavefront numbers of a round go through an empty asm: everything derived from them (LDS addresses,
    // the point slice) is then computed INSIDE the round instead of being hoisted in front of the loop and kept alive
    // across the fit, where the register allocator had to spill it (96-128 bytes of scratch per lane, and spill
    // stores are HBM writes: 125 MB per 10 000-pair launch).
    constexpr int R = 16;
    const int base = 0;
    int lane_r = lane, wave_r = wave;
    asm volatile("" : "+v"(lane_r), "+v"(wave_r));
    const int it = base + lane_r;
    if (wave_r == fit_wave) {
      int valid = 0;
      if (lane_r < R && it <= max_it) {
        uint32_t s0, s1, s2;
        if (draw_sample(L, P.seed, (uint32_t)it, P.max_sample_checks, (uint32_t)m, reinterpret_cast<const double*>(L.sums)[0],
                        s0, s1, s2)) {
          float coef[12];
          fit3(L, s0, s1, s2, coef);
          int lane_s = lane;                       // (addresses of the stores below: derived behind the fit)
          asm volatile("" : "+v"(lane_s));
#pragma unroll
          for (int k = 0; k < 12; ++k) L.hyp[k * 64 + lane_s] = coef[k];
          valid = 1;
        }
      }
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      hv[lane_s] = valid;
    }
    __syncthreads();
    SF_TRACE_MARK(P, pair, trace_base == 2 ? 18 : 21);   // first round: models parked
    {
      int lane_c = lane, wave_c = wave;
      asm volatile("" : "+v"(lane_c), "+v"(wave_c));
      const double thr2 = P.inlier_thr * P.inlier_thr;
      float thr2f = (float)thr2;                             // largest float strictly below thr^2
      if ((double)thr2f >= thr2) thr2f = __uint_as_float(__float_as_uint(thr2f) - 1u);
      const int slice = (m + NW - 1) / NW;                     // points counted by each wavefront
      const int i0 = min(m, wave_c * slice), i1 = min(m, i0 + slice);
      L.hyp_cnt[wave_c * 64 + lane_c] = count_round<16>(L, hv, i0, i1, lane_c, thr2f);
    }
    __syncthreads();
    SF_TRACE_MARK(P, pair, trace_base == 2 ? 19 : 22);   // ... inliers counted
    if (wave_r == fit_wave) {
      asm volatile("" : "+v"(lane_r));
      int tot = 0;
      const int h = lane_r & 15;
#pragma unroll
      for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) tot += L.hyp_cnt[w * 64 + g * 16 + h];
      const bool present = lane_r < R && it <= max_it;
      const int c = (present && hv[lane_r]) ? tot : -1;
      ScanState S;
      S.best = L.misc[12]; S.best_it = L.misc[13]; S.it = L.misc[14];
      S.k = reinterpret_cast<const double*>(L.sums)[1];
      const bool stop = replay_round(L, S, base, R, lane_r, c, present, max_it, P.adaptive_stop != 0,
                                     reinterpret_cast<const double*>(L.sums)[2],
                                     sfd::canon_log(1.0 - 0.99));
      if (lane_r == 0) {
        L.misc[12] = S.best; L.misc[13] = S.best_it; L.misc[14] = S.it;
        reinterpret_cast<double*>(L.sums)[1] = S.k;
        L.misc[0] = S.best_it;
        L.misc[1] = stop ? 1 : 0;
      }
    }
    __syncthreads();
    SF_TRACE_MARK(P, pair, trace_base == 2 ? 20 : 23);   // ... PCL's loop replayed
  }
  // The rounds behind the first (round 5): EVERY wavefront fits 64 hypotheses of its own -- iterations base + 64 w + lane
  // -- and counts them over all points itself, the model staying in the lane's registers; one wavefront then replays PCL's
  // loop over the 64 NW counts in iteration order (the same replay_round calls as with rounds of 64), and the lane that
  // owns a new best parks its model.  Rounds 2-4 had ONE wavefront fit 64 hypotheses while the others waited: with every
  // hypothesis evaluated (ransac_adaptive_stop = 0: the configuration read to the letter, 501 per pass) the fits were
  // three quarters of a chain.  Which iteration draws which sample does not depend on the round structure, so the
  // results are the same bytes.
  if (!L.misc[1]) {
    for (int base = 16; base <= max_it; base += 64 * NW) {
      int lane_r = lane, wave_r = wave;
      asm volatile("" : "+v"(lane_r), "+v"(wave_r));
      const int it = base + 64 * wave_r + lane_r;
      float coef[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) coef[k] = 0.f;
      int cnt = -1;                                    // (-1: no such iteration, or getSamples failed)
      if (it <= max_it) {
        uint32_t s0, s1, s2;
        if (draw_sample(L, P.seed, (uint32_t)it, P.max_sample_checks, (uint32_t)m, reinterpret_cast<const double*>(L.sums)[0],
                        s0, s1, s2)) {
          fit3(L, s0, s1, s2, coef);
          const double thr2 = P.inlier_thr * P.inlier_thr;
          float thr2f = (float)thr2;                           // largest float strictly below thr^2
          if ((double)thr2f >= thr2) thr2f = __uint_as_float(__float_as_uint(thr2f) - 1u);
          cnt = count_own(L, coef, m, thr2f);
        }
      }
      {
        int lane_s = lane, wave_s = wave;
        asm volatile("" : "+v"(lane_s), "+v"(wave_s));
        L.hyp_cnt[wave_s * 64 + lane_s] = cnt;
      }
      __syncthreads();
      if (wave_r == fit_wave) {
        asm volatile("" : "+v"(lane_r));
        ScanState S;
        S.best = L.misc[12]; S.best_it = L.misc[13]; S.it = L.misc[14];
        S.k = reinterpret_cast<const double*>(L.sums)[1];
        bool stop = false;
#pragma unroll 1
        for (int sub = 0; sub < NW && !stop; ++sub) {
          const int b = base + 64 * sub;
          if (b > max_it) break;                       // (the sequential loop's own end: no further round exists)
          const bool present = b + lane_r <= max_it;
          const int c = present ? L.hyp_cnt[sub * 64 + lane_r] : -1;
          stop = replay_round(L, S, b, 64, lane_r, c, present, max_it, P.adaptive_stop != 0,
                              reinterpret_cast<const double*>(L.sums)[2], sfd::canon_log(1.0 - 0.99), false);
        }
        if (lane_r == 0) {
          L.misc[12] = S.best; L.misc[13] = S.best_it; L.misc[14] = S.it;
          reinterpret_cast<double*>(L.sums)[1] = S.k;
          L.misc[0] = S.best_it;
          L.misc[1] = stop ? 1 : 0;
        }
      }
      __syncthreads();
      if (cnt >= 0 && L.misc[0] == it) {               // this lane's iteration is the new best: keep its model
#pragma unroll
        for (int k = 0; k < 12; ++k) L.best[k] = coef[k];
      }
      __syncthreads();
      if (L.misc[1]) break;
    }
  }
  // (same for everything behind the loop: addresses are re-derived from a laundered thread index instead of
  //  being shared with -- and kept alive since -- the gather in front of it)
  int tid_b = tid;
  asm volatile("" : "+v"(tid_b));
  if (P.dbg_stop == 2 || P.dbg_stop == 3) { if (tid_b == 0) write_null_pass(out, m); return; }   // diagnostic truncation
  SF_TRACE_MARK(P, pair, trace_base + 2);
  const int best_it = L.misc[0];
  if (best_it < 0) {
    if (tid_b == 0) write_null_pass(out, m);
    return;
  }

  // ---- winning model (parked in LDS by the scan) --------------------------------------------------
  const double thr = P.inlier_thr;
  const double thr2 = thr * thr;
  const double sigma = P.refine_sigma;
  SelCounts sc = select_within<NW>(L, m, L.best, thr2, L.mask_a, nullptr, sigma, thr, tid_b);
  int n_inl = sc.n;
  const uint8_t* inl = L.mask_a;     // the final inlier set (std::swap(inliers_, new_inliers) below)
  int n_last = n_inl;
  const float* model = L.best;

  if (P.dbg_stop == 4) { if (tid_b == 0) write_null_pass(out, m); return; }   // diagnostic truncation (SF_RANSAC_STOP)
  SF_TRACE_MARK(P, pair, trace_base + 3);

  // ---- refine loop (copy of pcl::SampleConsensus::refineModel inside rtabmap) ----------------------
  if (P.refine_iterations > 0) {
    double error_threshold = thr;
    int refine_iterations = 0;
    bool inlier_changed = false;
    uint8_t* prev = L.mask_a;
    uint8_t* neu = L.mask_b;
    int n_prev = n_inl, n_new = 0;
    for (int i = tid_b; i < m; i += NT) neu[i] = 0;
    if (tid_b < 12) L.bc[tid_b] = L.best[tid_b];            // new_model_coefficients = model_coefficients
    int n_sizes = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0;  // last four pushed sizes (z1 newest)
    do {
      if (n_prev >= 3) fit_masked<NW>(L, m, prev, n_prev, tid_b);
      z4 = z3; z3 = z2; z2 = z1; z1 = n_prev;
      ++n_sizes;
      __syncthreads();                                // L.bc (and the cleared `neu`) visible
      // membership changes are counted against `prev`, the set selected one round ago, while `neu` (the set of
      // two rounds ago) is overwritten
      sc = select_within<NW>(L, m, L.bc, error_threshold * error_threshold, neu, prev, sigma, thr, tid_b);
      n_new = sc.n;
      n_last = n_new;
      if (n_new == 0) {
        ++refine_iterations;
        if (refine_iterations >= P.refine_iterations) break;
        continue;
      }
      // error_threshold = min(thr, sigma * sqrt(variance)), variance = 2.1981 * median: the median is only needed
      // when it is one of the `low` members (see SelCounts)
      if ((n_new >> 1) < sc.low) {
        const double variance = variance_of<NW>(L, m, n_new, tid_b);
        const double sthr = sigma * sqrt(variance);
        error_threshold = thr < sthr ? thr : sthr;
      } else {
        error_threshold = thr;
      }
      inlier_changed = false;
      { uint8_t* t = prev; prev = neu; neu = t; int tn = n_prev; n_prev = n_new; n_new = tn; }
      if (n_new != n_prev) {
        if (n_sizes >= 4 && z1 == z3 && z2 == z4) break;  // oscillating
        inlier_changed = true;
        continue;
      }
      inlier_changed = sc.diff != 0;
    } while (inlier_changed && ++refine_iterations < P.refine_iterations);
    inl = neu;
    n_inl = n_new;
    model = L.bc;
  }

  if (P.dbg_stop == 5) { if (tid_b == 0) write_null_pass(out, m); return; }
  SF_TRACE_MARK(P, pair, trace_base + 4);
  if (n_inl >= 3) {
    if (mask_out)
      for (int i = tid; i < m; i += NT)
        if (inl[i]) mask_out[L.cidx[i] & 0xFFFFu] = 1;
    const double variance = variance_of<NW>(L, m, n_last, tid_b);
    if (tid_b == 0) {
      PassState ps;
#pragma unroll
      for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
      ps.is_null = 1;
      ps.matches = m;
      ps.pad = 0;
      ps.var = variance;
      ps.var_ang = variance;
      ps.inliers = n_inl;
      if (n_inl >= P.min_inliers) {
        double R9[9], t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) R9[3 * i + j] = (double)model[4 * i + j];
          t[i] = (double)model[4 * i + 3];
        }
        bool allz = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) ps.T[4 * i + j] = (float)R9[3 * j + i];
          ps.T[4 * i + 3] = (float)(-((R9[i] * t[0] + R9[3 + i] * t[1]) + R9[6 + i] * t[2]));
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) allz = allz && (ps.T[i] == 0.f);
        ps.is_null = allz ? 1 : 0;
        if (P.force_3dof && !allz) sfd::to3dof_canon(ps.T);     // :1141-1143
      }
      out = ps;
    }
  } else if (tid_b == 0) {
    write_null_pass(out, m);
  }
  // (:1192-1370, the two-view bundle adjustment of this estimate: k_ba_pass, a launch of its own -- k_verify.hip)
  SF_TRACE_MARK(P, pair, trace_base + 5);
}

// The end-of-pass to3DoF of myRegistration.cpp:269-276 and, for pass 1, the one its result meets as the guess of
// pass 2 (:245-248): `times` applications by the thread that wrote the pass state.
__device__ inline void pass_to3dof(PassState& ps, int times) {
  if (ps.is_null) return;
  for (int t = 0; t < times; ++t) sfd::to3dof_canon(ps.T);
}

template <int DIR>
__global__ void __launch_bounds__(SF_BLOCK, 4)
k_ransac(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
         const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
         const uint32_t* __restrict__ corr, const CorrHeader* __restrict__ hdr,
         PassState* __restrict__ pass, uint8_t* __restrict__ mask, int extra_3dof, DeviceParams P) {
  if ((int)blockIdx.x >= *counter) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = list[blockIdx.x];
  ransac_body<DIR>(st, pair, pair_from[pair], pair_to[pair], corr + (size_t)pair * st.kcap, hdr[pair].n_corr,
                       pass[pair], P, smem_raw, 2, mask ? mask + (size_t)pair * st.kcap : nullptr);
  if (extra_3dof && threadIdx.x == 0) pass_to3dof(pass[pair], extra_3dof);
}

// Vis/ForwardEstOnly = false: the two directions' estimates of a pass merged as myRegistrationVis.cpp:1155-1189
// (union of the inlier ids; the matches are the same ids in both directions) and :1376-1394 (inverse of the backward
// transform; interpolate(0.5) when both exist, covariance their mean).  One wavefront per pair.
__global__ void __launch_bounds__(64)
k_merge_directions(const int32_t* __restrict__ list, const int32_t* __restrict__ counter, PassState* __restrict__ fwd,
                   const PassState* __restrict__ back, const uint8_t* __restrict__ mask_f,
                   const uint8_t* __restrict__ mask_b, int kcap, int extra_3dof) {
  if ((int)blockIdx.x >= *counter) return;
  const int pair = list[blockIdx.x], lane = threadIdx.x;
  int uni = 0;
  for (int i = lane; i < kcap; i += 64) uni += (mask_f[(size_t)pair * kcap + i] | mask_b[(size_t)pair * kcap + i]) ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) uni += __shfl_xor(uni, off);
  if (lane != 0) return;
  PassState a = fwd[pair];
  const PassState b = back[pair];
  a.inliers = uni;
  a.matches = a.matches > b.matches ? a.matches : b.matches;
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
  pass_to3dof(a, extra_3dof);
  fwd[pair] = a;
}

}  // namespace

size_t sf_ransac_lds_bytes(int kcap, int iterations) {
  (void)iterations;   // (round 1 kept one count per iteration in LDS; a round's counts now live in lanes)
  return sf_ransac_lds_bytes_dev(kcap);
}

int sf_launch_ransac(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass) {
  if (n <= 0) return SF_OK;
  const bool ba = c->dparams.bundle_adjustment != 0;
  // (the adjustment is a launch of its own behind the estimate: sf_launch_ba_pass; both directions WITH the adjustment:
  //  plain estimates, then k_merge_directions_ba adjusts over the union)
  const size_t lds = (sf_ransac_lds_bytes(st.kcap, c->dparams.iterations) + 15) & ~(size_t)15;
  if (lds > 160 * 1024) return sf_fail(c, SF_ERANGE, "RANSAC workgroup needs %zu B of LDS (> 160 KiB)", lds);
  const bool bidir = c->dparams.bidirectional != 0;
  bool& attr = c->ransac_attr_set;
  if (!attr) {   // per handle = per device
    SF_HIP(c, hipFuncSetAttribute((const void*)k_ransac<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SF_HIP(c, hipFuncSetAttribute((const void*)k_ransac<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  const int32_t* list = (const int32_t*)(pass == 1 ? c->list1.p : c->list3.p);
  const int32_t* counter = counters + (pass == 1 ? 0 : 2);
  PassState* ps = (PassState*)(pass == 1 ? c->pass1.p : c->pass2.p);
  // Reg/Force3DoF: the end-of-pass application, and for pass 1 the one its result meets as the guess of pass 2
  const int end_3dof = c->dparams.force_3dof ? (pass == 1 ? 2 : 1) : 0;
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
    hipLaunchKernelGGL(kern, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to, list, counter,
                       (const uint32_t*)(pass == 1 ? c->corr1.p : c->corr2.p),
                       (const CorrHeader*)(pass == 1 ? c->hdr1.p : c->hdr2.p), out, mask, extra, c->dparams);
  };
  if (ba && !bidir) launch(k_ransac<0>, ps, mask_f, 0);     // (Reg/Force3DoF's end-of-pass application: behind the adjustment)
  else if (!bidir) launch(k_ransac<0>, ps, nullptr, end_3dof);
  else {
    launch(k_ransac<0>, ps, mask_f, 0);
    launch(k_ransac<1>, (PassState*)c->pass_back.p, mask_b, 0);
    if (ba) {
      const int rc = sf_launch_merge_directions_ba(c, st, d_from, d_to, n, pass, false, mask_f, mask_b);
      if (rc != SF_OK) return rc;
    } else
    hipLaunchKernelGGL(k_merge_directions, dim3(n), dim3(64), 0, c->stream, list, counter, ps,
                       (const PassState*)c->pass_back.p, (const uint8_t*)mask_f, (const uint8_t*)mask_b, st.kcap, end_3dof);
  }
  sf_prof_end(c, kid);
  SF_HIP(c, hipGetLastError());
  if (ba && !bidir) return sf_launch_ba_pass(c, st, d_from, d_to, n, pass, pass == 1 ? 1 : 3, mask_f, nullptr, false, nullptr);
  return SF_OK;
}
