// k_match.hip -- pass-1 GLOBAL matching: brute-force Hamming kNN (k=2) + NNDR + uniqueness.
//
// Replaces the dictionary round trip at myRegistrationVis.cpp:826-895 of the reference
// (VWDictionary::addNewWords in brute-force mode [upstream rtabmap] = cv::BFMatcher NORM_HAMMING
// knnMatch k=2, accept nearest id unless d1 > nndr*d2, keep ids occurring exactly once per side).
//
// CDNA4 mapping (one 256-thread workgroup per candidate pair):
//   * every lane keeps TWO "to" descriptors resident in VGPRs (8 or 16 dwords each);
//   * the "from" descriptors are wave-uniform, so they are fetched with SCALAR loads
//     (s_load_dwordx8/x16 through the scalar cache) and fed to v_xor_b32 as SGPR operands --
//     no LDS traffic and no per-lane address math in the K_from x K_to inner product;
//   * v_bcnt_u32_b32 accumulates the popcount; distance and index are packed into one 32-bit
//     key (dist << 16 | from_idx) so best/second-best tracking is 3 integer min/max ops and ties
//     resolve to the lowest index (BFMatcher order);
//   * uniqueness uses LDS counters; the id-ordered compaction uses wavefront ballots.
// The kernel is VALU-bound (~19 lane-ops per descriptor pair); HBM traffic is the two descriptor
// blocks, read once (the "to" block coalesced 16 B/lane, the "from" block via the scalar cache).
#include "sf_internal.hpp"

namespace {

// popcount(x) + acc in ONE VALU op; written as asm because LLVM's reassociation otherwise turns the
// accumulate chain into `v_bcnt x, 0` + `v_add3` trees (+3 VALU ops per 256-bit descriptor pair).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

// Scan all "from" descriptors (wave-uniform -> scalar loads, SGPR operands) against the NQ "to"
// descriptors this lane keeps in VGPRs; track best / second-best keys (dist << 16 | from_idx).
template <int W, int NQ>
__device__ __forceinline__ void knn2_scan(const uint32_t* __restrict__ dF, int Kf, const uint32_t (&q)[NQ][W],
                                          uint32_t (&k1)[NQ], uint32_t (&k2)[NQ]) {
#pragma unroll 4
  for (int f = 0; f < Kf; ++f) {
    const uint32_t* r = dF + (size_t)f * W;
    uint32_t x[W];
#pragma unroll
    for (int c = 0; c < W; ++c) x[c] = r[c];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      uint32_t d = 0;
#pragma unroll
      for (int c = 0; c < W; ++c) d = bcnt_acc(x[c] ^ q[j][c], d);
      const uint32_t key = (d << 16) | (uint32_t)f;
      k2[j] = min(max(key, k1[j]), k2[j]);
      k1[j] = min(k1[j], key);
    }
  }
}

template <int W>
__device__ __forceinline__ void load_desc(const uint32_t* __restrict__ base, int row, bool valid, uint32_t (&q)[W]) {
  const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)(valid ? row : 0) * W);
#pragma unroll
  for (int c = 0; c < W / 4; ++c) {
    uint4 v = p[c];
    q[4 * c + 0] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
  }
}

// W  : dwords per descriptor (8 / 16)
// NQ : "to" descriptors resident per lane
// NT : threads per workgroup (one workgroup per candidate pair)
template <int W, int NQ, int NT>
__global__ void __launch_bounds__(NT)
k_match_global(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
               float nndr, int min_inliers, int est, uint32_t* __restrict__ corr, CorrHeader* __restrict__ hdr,
               PassState* __restrict__ pass, int32_t* __restrict__ list, int32_t* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  constexpr int NW = NT / 64;
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  if ((unsigned)sF >= (unsigned)st.n_slots || (unsigned)sT >= (unsigned)st.n_slots) {
    // caller-provided slot outside the store: report a failed estimation, touch nothing else
    if (tid == 0) {
      CorrHeader h = {0, 0, 0, 0};
      hdr[pair] = h;
      PassState ps;
#pragma unroll
      for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
      ps.var = 1.0; ps.var_ang = 1.0; ps.is_null = 1; ps.inliers = 0; ps.matches = 0; ps.pad = 0;
      pass[pair] = ps;
    }
    return;
  }
  const int4 mF = st.meta[sF], mT = st.meta[sT];
  const int Kf = mF.x, Kt = mT.x;
  const uint32_t* dF = st.desc + (size_t)sF * kcap * W;
  const uint32_t* dT = st.desc + (size_t)sT * kcap * W;

  int* cnt = smem;               // [kcap] "to" rows that matched each "from" word
  int* owner = smem + kcap;      // [kcap] the matching "to" row (meaningful when cnt == 1)
  int* misc = smem + 2 * kcap;   // [16]   0: rejected "to" rows, 2: finite corr, 4..7 wave totals

  for (int i = tid; i < Kf; i += NT) cnt[i] = 0;
  if (tid < 16) misc[tid] = 0;
  __syncthreads();

  int rejected = 0;
  if (Kf > 0) {
    for (int base = 0; base < Kt; base += NQ * NT) {
      uint32_t q[NQ][W], k1[NQ], k2[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int t = base + j * NT + tid;
        load_desc<W>(dT, t, t < Kt, q[j]);
        k1[j] = 0xFFFFFFFFu;
        k2[j] = 0xFFFFFFFFu;
      }
      // wave-uniform: how many of this wave's NQ row groups hold at least one valid row?
      const int wave_rows = Kt - base - (tid & ~63);
      const int groups = wave_rows <= 0 ? 0 : min(NQ, (wave_rows + NT - 1) / NT);
      if (groups == NQ) {
        knn2_scan<W, NQ>(dF, Kf, q, k1, k2);
      } else if (groups > 0) {
        // ragged tail: scan only the populated groups (compile-time indices keep q[] in registers)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j < groups) {
            uint32_t qq[1][W], a1[1] = {0xFFFFFFFFu}, a2[1] = {0xFFFFFFFFu};
#pragma unroll
            for (int c = 0; c < W; ++c) qq[0][c] = q[j][c];
            knn2_scan<W, 1>(dF, Kf, qq, a1, a2);
            k1[j] = a1[0];
            k2[j] = a2[0];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int t = base + j * NT + tid;
        if (t < Kt) {
          const bool acc = (Kf >= 2) && !((float)(k1[j] >> 16) > nndr * (float)(k2[j] >> 16));
          if (acc) {
            const int f = (int)(k1[j] & 0xFFFFu);
            atomicAdd(&cnt[f], 1);
            owner[f] = t;
          } else {
            ++rejected;
          }
        }
      }
    }
  }
  // wave-reduce the rejected count, one LDS atomic per wave
  for (int off = 32; off >= 1; off >>= 1) rejected += __shfl_xor(rejected, off);
  if (lane == 0 && rejected) atomicAdd(&misc[0], rejected);
  __syncthreads();

  // id-ordered compaction of the "from" words matched by exactly one "to" row
  uint32_t* out = corr + (size_t)pair * kcap;
  int running = 0;
  for (int base = 0; base < Kf; base += NT) {
    const int f = base + tid;
    const bool flag = (f < Kf) && (cnt[f] == 1);
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int c = misc[4 + w];
      if (w < wave) woff += c;
      total += c;
    }
    if (flag) out[running + woff + before] = (uint32_t)f | ((uint32_t)owner[f] << 16);
    running += total;
    __syncthreads();
  }
  const int n_corr = running;

  // Gate of the 3D->3D estimation (myRegistrationVis.cpp:928,1117-1118) and, when RANSAC will not
  // run, the `matches` count util3d::findCorrespondences would still report.
  const int unique_to = (Kf > 0 && Kt > 0) ? misc[0] + n_corr : 0;
  const int words_from = (Kf > 0 && mF.y > 0) ? Kf : 0;
  const int words_to = (mT.y > 0) ? unique_to : 0;
  // est: 0 = 3D->3D gate (:1117-1118); 1 = PnP gate (:1070-1071, 2D words of the "to" frame);
  //      2 = PnP without a calibrated camera (:1059-1065): the estimation never runs
  const bool motion = est == 0 ? (unique_to > 0 && words_from >= min_inliers && words_to >= min_inliers)
                               : (est == 1 && unique_to > 0 && words_from >= min_inliers && unique_to >= min_inliers);
  const bool survivor = motion && n_corr >= min_inliers && n_corr >= (est == 0 ? 3 : 4);
  if (motion && !survivor) {
    const float* xF = st.xyz + (size_t)sF * kcap * 3;
    const float* xT = st.xyz + (size_t)sT * kcap * 3;
    for (int i = tid; i < n_corr; i += NT) {
      uint32_t c = out[i];
      const float* a = xF + 3 * (c & 0xFFFFu);
      const float* b = xT + 3 * (c >> 16);
      bool ok = isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2]);
      if (est == 0)   // findCorrespondences (3D-3D) also needs the "to" point and drops zero points
        ok = ok && isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f) &&
             (b[0] != 0.f || b[1] != 0.f || b[2] != 0.f);
      if (ok) atomicAdd(&misc[2], 1);
    }
    __syncthreads();
  }
  if (tid == 0) {
    CorrHeader h;
    h.n_corr = n_corr;
    h.words_from = words_from;
    h.words_to = words_to;
    h.words_to_2d = unique_to;
    hdr[pair] = h;
    PassState ps;
#pragma unroll
    for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
    ps.var = 1.0; ps.var_ang = 1.0;
    ps.is_null = 1;
    ps.inliers = 0;
    ps.matches = (motion && !survivor) ? misc[2] : 0;
    ps.pad = 0;
    pass[pair] = ps;
    if (survivor) {
      int pos = atomicAdd(counter, 1);
      list[pos] = pair;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Variant 2 ("LDS + u16"), shaped by the measured gfx950 VALU issue rates (tools/ubench/valu_rate.hip:
// v_xor_b32 VGPR,VGPR / v_min_u16 / v_max_u16 issue at full rate; v_bcnt_u32_b32, every 32-bit
// min/max, v_lshl_or_b32 and any VALU op with an SGPR operand at HALF rate):
//   * the "from" block is staged once into LDS and broadcast-read into VGPRs (uniform address: one
//     LDS cycle group, no conflicts), so the xor is VGPR x VGPR;
//   * best and second-best are tracked as 16-bit DISTANCES only (3 full-rate ops, no key packing);
//   * which "from" row holds the minimum is recovered afterwards, for accepted lanes only, by
//     re-scanning the 16-row chunk in which the running minimum last decreased (first row whose
//     distance equals the minimum = lowest index, the tie rule of the packed-key variant).
__device__ __forceinline__ uint32_t min_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

constexpr int MATCH_CH = 16;   // rows per index-recovery chunk

template <int W, int NQ>
__device__ __forceinline__ void knn2_step_lds(const uint32_t* fromD, int f, const uint32_t (&q)[NQ][W],
                                              uint32_t (&d1)[NQ], uint32_t (&d2)[NQ]) {
  const uint4* r = reinterpret_cast<const uint4*>(fromD + (size_t)f * W);   // same address in every lane
  uint32_t x[W];
#pragma unroll
  for (int c = 0; c < W / 4; ++c) {
    const uint4 v = r[c];
    x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    uint32_t d = 0;
#pragma unroll
    for (int c = 0; c < W; ++c) d = bcnt_acc(x[c] ^ q[j][c], d);
    d2[j] = min_u16(d2[j], max_u16(d, d1[j]));
    d1[j] = min_u16(d1[j], d);
  }
}

template <int W, int NQ>
__device__ __forceinline__ void knn2_scan_lds(const uint32_t* fromD, int Kf, const uint32_t (&q)[NQ][W],
                                              uint32_t (&d1)[NQ], uint32_t (&d2)[NQ], uint32_t (&chunk)[NQ]) {
  int c0 = 0;
  // full chunks: compile-time trip count, so the LDS reads of several rows are issued ahead
  for (; c0 + MATCH_CH <= Kf; c0 += MATCH_CH) {
    uint32_t prev[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) prev[j] = d1[j];
#pragma unroll 4
    for (int u = 0; u < MATCH_CH; ++u) knn2_step_lds<W, NQ>(fromD, c0 + u, q, d1, d2);
#pragma unroll
    for (int j = 0; j < NQ; ++j) chunk[j] = ((d1[j] & 0xFFFFu) != (prev[j] & 0xFFFFu)) ? (uint32_t)c0 : chunk[j];
  }
  if (c0 < Kf) {   // ragged last chunk
    uint32_t prev[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) prev[j] = d1[j];
    for (int f = c0; f < Kf; ++f) knn2_step_lds<W, NQ>(fromD, f, q, d1, d2);
#pragma unroll
    for (int j = 0; j < NQ; ++j) chunk[j] = ((d1[j] & 0xFFFFu) != (prev[j] & 0xFFFFu)) ? (uint32_t)c0 : chunk[j];
  }
}

// Body of the matching stage for ONE pair (the calling workgroup); `smem` is the workgroup's dynamic
// LDS.  Returns whether the pair goes on to motion estimation (block-uniform).  With list == nullptr
// the pair is not appended to a work list (fused pipeline, k_verify.hip).
template <int W, int NQ, int NT>
__device__ __forceinline__ bool match_v2_body(const StoreView& st, int pair, const int32_t* __restrict__ pair_from,
                                              const int32_t* __restrict__ pair_to, float nndr, int min_inliers, int est,
                                              uint32_t* __restrict__ corr, CorrHeader* __restrict__ hdr,
                                              PassState* __restrict__ pass, int32_t* __restrict__ list,
                                              int32_t* __restrict__ counter, int* smem) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  if ((unsigned)sF >= (unsigned)st.n_slots || (unsigned)sT >= (unsigned)st.n_slots) {
    if (tid == 0) {
      CorrHeader h = {0, 0, 0, 0};
      hdr[pair] = h;
      PassState ps;
#pragma unroll
      for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
      ps.var = 1.0; ps.var_ang = 1.0; ps.is_null = 1; ps.inliers = 0; ps.matches = 0; ps.pad = 0;
      pass[pair] = ps;
    }
    return false;
  }
  const int4 mF = st.meta[sF], mT = st.meta[sT];
  const int Kf = mF.x, Kt = mT.x;
  const uint32_t* dF = st.desc + (size_t)sF * kcap * W;
  const uint32_t* dT = st.desc + (size_t)sT * kcap * W;

  uint32_t* fromD = reinterpret_cast<uint32_t*>(smem);   // [kcap * W] staged "from" descriptors
  int* cnt = smem + kcap * W;                              // [kcap]
  int* owner = cnt + kcap;                                 // [kcap]
  int* misc = owner + kcap;                                // [16]

  {
    const uint4* src = reinterpret_cast<const uint4*>(dF);
    uint4* dst = reinterpret_cast<uint4*>(fromD);
    for (int i = tid; i < Kf * (W / 4); i += NT) dst[i] = src[i];
  }
  for (int i = tid; i < Kf; i += NT) cnt[i] = 0;
  if (tid < 16) misc[tid] = 0;
  __syncthreads();

  int rejected = 0;
  if (Kf > 0) {
    for (int base = 0; base < Kt; base += NQ * NT) {
      uint32_t q[NQ][W], d1[NQ], d2[NQ], chunk[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int t = base + j * NT + tid;
        load_desc<W>(dT, t, t < Kt, q[j]);
        d1[j] = 0xFFFFu;
        d2[j] = 0xFFFFu;
        chunk[j] = 0;
      }
      const int wave_rows = Kt - base - (tid & ~63);
      const int groups = wave_rows <= 0 ? 0 : min(NQ, (wave_rows + NT - 1) / NT);
      if (groups == NQ) {
        knn2_scan_lds<W, NQ>(fromD, Kf, q, d1, d2, chunk);
      } else if (groups > 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j < groups) {
            uint32_t qq[1][W], a1[1] = {0xFFFFu}, a2[1] = {0xFFFFu}, ch[1] = {0};
#pragma unroll
            for (int c = 0; c < W; ++c) qq[0][c] = q[j][c];
            knn2_scan_lds<W, 1>(fromD, Kf, qq, a1, a2, ch);
            d1[j] = a1[0]; d2[j] = a2[0]; chunk[j] = ch[0];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int t = base + j * NT + tid;
        if (t < Kt) {
          const uint32_t b1 = d1[j] & 0xFFFFu, b2 = d2[j] & 0xFFFFu;
          const bool acc = (Kf >= 2) && !((float)b1 > nndr * (float)b2);
          if (acc) {
            // index recovery inside the chunk where the minimum was reached
            int f = (int)chunk[j];
            const int fend = min(f + MATCH_CH, Kf);
            for (; f < fend; ++f) {
              const uint32_t* r = fromD + (size_t)f * W;
              uint32_t d = 0;
#pragma unroll
              for (int c = 0; c < W; ++c) d += __popc(r[c] ^ q[j][c]);
              if (d == b1) break;
            }
            atomicAdd(&cnt[f], 1);
            owner[f] = t;
          } else {
            ++rejected;
          }
        }
      }
    }
  }
  for (int off = 32; off >= 1; off >>= 1) rejected += __shfl_xor(rejected, off);
  if (lane == 0 && rejected) atomicAdd(&misc[0], rejected);
  __syncthreads();

  uint32_t* out = corr + (size_t)pair * kcap;
  int running = 0;
  for (int base = 0; base < Kf; base += NT) {
    const int f = base + tid;
    const bool flag = (f < Kf) && (cnt[f] == 1);
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[4 + wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int c = misc[4 + w];
      if (w < wave) woff += c;
      total += c;
    }
    if (flag) out[running + woff + before] = (uint32_t)f | ((uint32_t)owner[f] << 16);
    running += total;
    __syncthreads();
  }
  const int n_corr = running;

  const int unique_to = (Kf > 0 && Kt > 0) ? misc[0] + n_corr : 0;
  const int words_from = (Kf > 0 && mF.y > 0) ? Kf : 0;
  const int words_to = (mT.y > 0) ? unique_to : 0;
  // est: 0 = 3D->3D gate (:1117-1118); 1 = PnP gate (:1070-1071, 2D words of the "to" frame);
  //      2 = PnP without a calibrated camera (:1059-1065): the estimation never runs
  const bool motion = est == 0 ? (unique_to > 0 && words_from >= min_inliers && words_to >= min_inliers)
                               : (est == 1 && unique_to > 0 && words_from >= min_inliers && unique_to >= min_inliers);
  const bool survivor = motion && n_corr >= min_inliers && n_corr >= (est == 0 ? 3 : 4);
  if (motion && !survivor) {
    const float* xF = st.xyz + (size_t)sF * kcap * 3;
    const float* xT = st.xyz + (size_t)sT * kcap * 3;
    for (int i = tid; i < n_corr; i += NT) {
      uint32_t c = out[i];
      const float* a = xF + 3 * (c & 0xFFFFu);
      const float* b = xT + 3 * (c >> 16);
      bool ok = isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2]);
      if (est == 0)   // findCorrespondences (3D-3D) also needs the "to" point and drops zero points
        ok = ok && isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f) &&
             (b[0] != 0.f || b[1] != 0.f || b[2] != 0.f);
      if (ok) atomicAdd(&misc[2], 1);
    }
    __syncthreads();
  }
  if (tid == 0) {
    CorrHeader h;
    h.n_corr = n_corr;
    h.words_from = words_from;
    h.words_to = words_to;
    h.words_to_2d = unique_to;
    hdr[pair] = h;
    PassState ps;
#pragma unroll
    for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
    ps.var = 1.0; ps.var_ang = 1.0;
    ps.is_null = 1;
    ps.inliers = 0;
    ps.matches = (motion && !survivor) ? misc[2] : 0;
    ps.pad = 0;
    pass[pair] = ps;
    if (survivor && list) {
      int pos = atomicAdd(counter, 1);
      list[pos] = pair;
    }
  }
  return survivor;
}

template <int W, int NQ, int NT>
__global__ void __launch_bounds__(NT)
k_match_global_v2(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
                  float nndr, int min_inliers, int est, uint32_t* __restrict__ corr, CorrHeader* __restrict__ hdr,
                  PassState* __restrict__ pass, int32_t* __restrict__ list, int32_t* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  match_v2_body<W, NQ, NT>(st, (int)blockIdx.x, pair_from, pair_to, nndr, min_inliers, est, corr, hdr, pass, list,
                           counter, smem);
}

template <int W, int NQ, int NT>
void launch_match_v2(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  size_t lds = (size_t)(st.kcap * W + 2 * st.kcap + 16) * sizeof(int);
  if (const char* v = getenv("SF_MATCH_LDS_PAD")) lds += (size_t)atoi(v);   // occupancy experiment (diagnostic)
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_match_global_v2<W, NQ, NT>), dim3(n), dim3(NT), lds, c->stream, st, d_from, d_to,
                     c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p,
                     (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
}

template <int W, int NQ, int NT>
void launch_match(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  const size_t lds = (size_t)(2 * st.kcap + 16) * sizeof(int);
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_match_global<W, NQ, NT>), dim3(n), dim3(NT), lds, c->stream, st, d_from, d_to,
                     c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p,
                     (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
}

}  // namespace

int sf_launch_match_global(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  if (n <= 0) return SF_OK;
  // geometry: variant = NQ * 1000 + NT (tunable through SF_MATCH_VARIANT for A/B runs)
  int variant = c->match_variant;
  if (variant == 0) {
    // default: LDS + u16 variant while the staged "from" block keeps >= 2 workgroups per CU
    const size_t lds_v2 = (size_t)(st.kcap * st.w + 2 * st.kcap + 16) * sizeof(int);
    variant = lds_v2 <= 64 * 1024 ? 12256 : 2256;
  }
  sf_prof_begin(c, SF_K_MATCH);
#define SF_CASE(NQ_, NT_)                                                        \
  case NQ_ * 1000 + NT_:                                                         \
    if (st.w == 8) launch_match<8, NQ_, NT_>(c, st, d_from, d_to, n);            \
    else launch_match<16, NQ_, NT_>(c, st, d_from, d_to, n);                     \
    break;
  switch (variant) {
    SF_CASE(1, 256)
    SF_CASE(2, 256)
    SF_CASE(4, 128)
    SF_CASE(2, 128)
    SF_CASE(4, 256)
    SF_CASE(8, 64)
    SF_CASE(4, 64)
#define SF_CASE2(NQ_, NT_)                                                          \
  case 10000 + NQ_ * 1000 + NT_:                                                    \
    if (st.w == 8) launch_match_v2<8, NQ_, NT_>(c, st, d_from, d_to, n);            \
    else launch_match_v2<16, NQ_, NT_>(c, st, d_from, d_to, n);                     \
    break;
    SF_CASE2(2, 256)
    SF_CASE2(1, 256)
    SF_CASE2(4, 128)
    SF_CASE2(2, 128)
#undef SF_CASE2
    default:
      sf_prof_end(c, SF_K_MATCH);
      return sf_fail(c, SF_EINVAL, "unknown match kernel variant %d", variant);
  }
#undef SF_CASE
  sf_prof_end(c, SF_K_MATCH);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
