// k_match.hip -- pass-1 GLOBAL matching: brute-force Hamming kNN (k=2) + NNDR + uniqueness.
//
// Replaces the dictionary round trip at myRegistrationVis.cpp:826-895 of the reference
// (VWDictionary::addNewWords in brute-force mode [upstream rtabmap] = cv::BFMatcher NORM_HAMMING
// knnMatch k=2, accept nearest id unless d1 > nndr*d2, keep ids occurring exactly once per side).
//
// Three formulations of the same integers, all one 256-thread workgroup per candidate pair:
//   variant 3 (default, NQ == 0, further down): the K_from x K_to distance table as an exact +-1 product
//             on the fp4 matrix cores; variant 2 ("LDS + u16", SF_MATCH_MFMA=0): xor + popcount on the VALU
//             with the "from" block in LDS; variant 1 (frames too large for the LDS copy): below.
// Variant 1, CDNA4 mapping:
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
#include <type_traits>

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

// Float32 descriptor rows (desc_type 1: SURF / SIFT, W = dimensions): squared L2 distance of every "from" row to the ONE
// "to" row this lane keeps in VGPRs, accumulated in float32 in dimension order, multiply and add unfused (the
// translation unit is built with -ffp-contract=off) -- the oracle's loop, bit for bit.  Strict comparisons: ties keep
// the lower "from" index (BFMatcher order); a NaN distance is never taken.
template <int W>
__device__ __forceinline__ void knn2_scan_l2(const uint32_t* __restrict__ dF, int Kf, const uint32_t (&q)[W], float& d1,
                                             float& d2, int& i1) {
#pragma unroll 1
  for (int f = 0; f < Kf; ++f) {
    const uint32_t* r = dF + (size_t)f * W;         // wave-uniform: scalar loads
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const float d = __uint_as_float(q[c]) - __uint_as_float(r[c]);
      s = s + d * d;
    }
    if (s < d1) { d2 = d1; d1 = s; i1 = f; }
    else if (s < d2) { d2 = s; }
  }
}

// The same scan for rows of more than 64 dimensions (SIFT: 128) without keeping the whole "to" row in registers: the
// 128 registers of the resident row left two wavefronts per SIMD to cover the scalar loads (9.8 ms per 2 048 pairs of
// K = 500 against an issue bound of 6.2, profiles/r05zb_float_descriptors.txt).  Here a lane walks the "from" rows in blocks
// of FB: the first 64 dimensions of the block's rows against the first half of its "to" row, then the second half -- one
// 64-register half resident at a time, re-read from L2 per block (256 B per lane against 3 x 64 x FB vector instructions),
// FB partial sums carried between the halves.  Per (from, to) pair the sum still runs over the dimensions in order: the
// same operations on the same values, the same bits.
template <int W>
__device__ __forceinline__ void knn2_scan_l2_halves(const uint32_t* __restrict__ dF, int Kf,
                                                    const uint32_t* __restrict__ qrow, float& d1, float& d2, int& i1) {
  static_assert(W == 128, "two halves of 64 dimensions");
  constexpr int H = 64, FB = 16;
#pragma unroll 1
  for (int f0 = 0; f0 < Kf; f0 += FB) {
    float s[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j) s[j] = 0.f;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      uint32_t q[H];
      {
        const uint4* p = reinterpret_cast<const uint4*>(qrow + half * H);
#pragma unroll
        for (int c = 0; c < H / 4; ++c) {
          const uint4 v = p[c];
          q[4 * c] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
        }
      }
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        if (f0 + j < Kf) {                                       // wave-uniform
          const uint32_t* r = dF + (size_t)(f0 + j) * W + half * H;   // wave-uniform: scalar loads
          float a = s[j];
#pragma unroll
          for (int c = 0; c < H; ++c) {
            const float d = __uint_as_float(q[c]) - __uint_as_float(r[c]);
            a = a + d * d;
          }
          s[j] = a;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      if (f0 + j < Kf) {
        if (s[j] < d1) { d2 = d1; d1 = s[j]; i1 = f0 + j; }
        else if (s[j] < d2) { d2 = s[j]; }
      }
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
// L2 : float32 descriptor rows of W dimensions (squared L2, NQ = 1) instead of binary rows of W dwords (Hamming)
template <int W, int NQ, int NT, bool L2 = false>
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
  if constexpr (L2) {
    static_assert(!L2 || NQ == 1, "one resident float row per lane");
    if (Kf > 0) {
      for (int base = 0; base < Kt; base += NT) {
        const int t = base + tid;
        float d1 = __int_as_float(0x7F800000), d2 = __int_as_float(0x7F800000);
        int i1 = -1;
        if constexpr (W > 64) {
          if (Kt - base - (tid & ~63) > 0)        // (wave-uniform: this wavefront holds at least one valid row)
            knn2_scan_l2_halves<W>(dF, Kf, dT + (size_t)(t < Kt ? t : 0) * W, d1, d2, i1);
        } else {
          uint32_t q[W];
          load_desc<W>(dT, t, t < Kt, q);
          if (Kt - base - (tid & ~63) > 0)
            knn2_scan_l2<W>(dF, Kf, q, d1, d2, i1);
        }
        if (t < Kt) {
          const bool acc = (Kf >= 2) && i1 >= 0 && !(d1 > nndr * d2);
          if (acc) {
            atomicAdd(&cnt[i1], 1);
            owner[i1] = t;
          } else {
            ++rejected;
          }
        }
      }
    }
  } else
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
  //      3 = PnP, both directions (Vis/ForwardEstOnly = false): either gate; every pair with a correspondence goes on,
  //          the estimates count their own matches (the union of the two is not this kernel's to know)
  const bool motion = est == 0 ? (unique_to > 0 && words_from >= min_inliers && words_to >= min_inliers)
                    : est == 1 ? (unique_to > 0 && words_from >= min_inliers && unique_to >= min_inliers)
                               : (est == 3 && unique_to > 0 && ((words_from >= min_inliers && unique_to >= min_inliers) ||
                                                                (words_to >= min_inliers && Kf >= min_inliers)));
  const bool survivor = est == 3 ? (motion && n_corr > 0) : (motion && n_corr >= min_inliers && n_corr >= (est == 0 ? 3 : 4));
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

// ------------------------------------------------------------------------------------------------
// Variant 3 ("fp4 matrix cores", NQ == 0): the K_from x K_to Hamming table as a matrix product.
// With every descriptor bit b mapped to the fp4 (E2M1) value 1 - 2b, the dot product of two rows is
// (bits - 2 * hamming); the coding actually used (fp4_spread_from / fp4_spread_to below) gives the same number
// minus a constant of the "to" row for 4 instead of 7 VALU ops per streamed dword.  The products are exact in
// fp4 and their sums (|.| <= 1024) are exact in the f32 accumulator, so the distances are the integers the VALU
// variants compute.  v_mfma_f32_32x32x64_f8f6f4
// does 32 x 32 x 64 of them per instruction (tools/ubench/mfma_fp4_hamming.hip pins operand layout and
// exactness on the device); the order of the 64 bits inside one instruction is irrelevant as long as
// both operands use the same one, so a lane simply spreads the raw dwords it loaded.
//   * "from" rows = matrix rows (A, re-read from the LDS copy per 32-row tile), "to" rows = columns
//     (B, kept in VGPRs for the whole scan); a wavefront owns 32-column tiles, two at a time;
//   * the accumulator starts at -(from row index)/2048 instead of 0, so one f32 per (from, to) cell
//     orders by distance first (steps of 2) and by the LOWER from index second (fraction < 1): the
//     BFMatcher tie rule with no separate index bookkeeping;
//   * best / second best per column: v_max_f32 + v_med3_f32 per cell, then one cross-half merge.
typedef int mf_v8i __attribute__((ext_vector_type(8)));
typedef float mf_v16f __attribute__((ext_vector_type(16)));
constexpr float MF_FR = 1.f / 2048.f;        // index fraction (kcap <= 2048 rows on this path)
constexpr int MF_MAX_ROWS = 2048;

// 32 descriptor bits -> 32 fp4 (E2M1) values, 4 dwords of nibbles.  The streamed operand (the "from" rows, spread
// once per tile by every wavefront) must be cheap, so its bits stay where they are: dword k of the result keeps
// the bits at nibble position 3 - k of the raw dword,
//   position 3: sign bit of 1.0          -> +1 / -1          (x & 0x8888.. | 0x2222..)
//   position 2: exponent bit, code 0100  ->  0 / 2.0         (x & 0x4444..)
//   position 1: exponent bit, code 0010  ->  0 / 1.0         (x & 0x2222..)
//   position 0: mantissa bit, code 0001  ->  0 / 0.5         (x & 0x1111..)
// 4 VALU ops per raw dword (the all-sign form ((x << k) & 0x8888..) | 0x2222.. costs 7).  With t = +1 / -1 for a
// bit 0 / 1, an element of the last three classes is a = d (1 - tx), d = 1, 1/2, 1/4; the resident operand (the
// "to" rows, spread once per pass) answers with b = -ty / d = -+1, -+2, -+4 (codes 0x2, 0x4, 0x6 under the sign),
// so a b = tx ty - ty, and the sign class contributes tx ty directly:
//   dot = (bits - 2 hamming) - sum_{classes 2,1,0} ty = (bits - 2 hamming) - (3 bits / 4 - 2 popc(y & 0x7777..)).
// The second term is a constant of the "to" row: it shifts every score of a column alike (best / second best and
// the tie rule are untouched) and is taken out when the distances are decoded.  Products and sums are exact.
// m88 / c22 hold 0x88888888 / 0x22222222 in VGPRs the compiler cannot see through (knn2_mfma): a VOP3 cannot
// encode a literal on gfx9, so v_and_or_b32 needs them in registers.  No asm on this path: the results are MFMA
// operands and the hazard recogniser must see the instructions that write them.
__device__ __forceinline__ mf_v8i fp4_spread_from(uint32_t x, uint32_t m88, uint32_t c22) {
  mf_v8i o = {0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (int)((x & m88) | c22);
  o[1] = (int)(x & 0x44444444u);
  o[2] = (int)(x & 0x22222222u);
  o[3] = (int)(x & 0x11111111u);
  return o;
}
__device__ __forceinline__ mf_v8i fp4_spread_to(uint32_t y, uint32_t m88, uint32_t c22) {
  mf_v8i o = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t n = ~y;
  o[0] = (int)((y & m88) | c22);
  o[1] = (int)(((n << 1) & m88) | c22);
  o[2] = (int)(((n << 2) & m88) | 0x44444444u);
  o[3] = (int)(((n << 3) & m88) | 0x66666666u);
  return o;
}

// Dynamic LDS of the LDS-staged matching body: the staged "from" descriptors, the per-row counters / owners and 16 words.
// The MFMA scan requests a tile's rows one tile ahead without clamping the row number, i.e. up to 32 rows past the
// staged block: the words behind it must exist (they are the counters, or padding when kcap is small).
static inline size_t sf_match_lds_bytes(int kcap, int w) {
  const int behind = 2 * kcap + 16;
  return (size_t)(kcap * w + (behind > 32 * w ? behind : 32 * w)) * sizeof(int);
}

template <int KS>
__device__ __forceinline__ void load_raw(const uint32_t* p, uint32_t (&raw)[KS]) {
#pragma unroll
  for (int c = 0; c < KS / 4; ++c) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[c];
    raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
  }
}

// (b, s) <- the two largest of {b, s, v[0..15]} (b >= s on entry and exit), 20 VALU ops:
//   per pair (v0, v1):  x = med3(b, v0, v1) is the second largest of {b, v0, v1}, b' = max3(b, v0, v1),
//   and the second largest of {b, s, v0, v1} is max(s, x) -- so two pairs cost 2 med3 + 3 max3.
// Written as asm because fmaxf() costs a canonicalising self-max per operand; the FIRST read of the
// accumulator is left to the compiler (a builtin), which places the MFMA-result wait states in front of
// it -- the hazard recogniser does not look inside an asm statement.
__device__ __forceinline__ void top2_update16(const mf_v16f& v, float& b, float& s) {
  const float x0 = __builtin_amdgcn_fmed3f(b, v[0], v[1]);
  float ta, tb;
  asm("v_max3_f32 %0, %0, %4, %5\n\t"
      "v_med3_f32 %2, %0, %6, %7\n\t"
      "v_max3_f32 %0, %0, %6, %7\n\t"
      "v_max3_f32 %1, %1, %20, %2\n\t"
      "v_med3_f32 %2, %0, %8, %9\n\t"
      "v_max3_f32 %0, %0, %8, %9\n\t"
      "v_med3_f32 %3, %0, %10, %11\n\t"
      "v_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %1, %1, %2, %3\n\t"
      "v_med3_f32 %2, %0, %12, %13\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\t"
      "v_med3_f32 %3, %0, %14, %15\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\t"
      "v_max3_f32 %1, %1, %2, %3\n\t"
      "v_med3_f32 %2, %0, %16, %17\n\t"
      "v_max3_f32 %0, %0, %16, %17\n\t"
      "v_med3_f32 %3, %0, %18, %19\n\t"
      "v_max3_f32 %0, %0, %18, %19\n\t"
      "v_max3_f32 %1, %1, %2, %3"
      : "+v"(b), "+v"(s), "=&v"(ta), "=&v"(tb)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
        "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(x0));
}

// One 32-row "from" tile against the NTL resident "to" tiles of this wavefront.
// `raw` holds this tile's rows, requested one tile ahead (an LDS round trip, ~100 cycles, no longer opens every
// iteration of a wavefront's scan: k_verify_fused 0.469 -> 0.457 ms per 10 000 pairs); the next tile's are requested
// here, behind the spread that consumes these.
template <int W, int NTL, bool TAIL>
__device__ __forceinline__ void knn2_mfma_tile(const uint32_t* fromD, int Kf, int mt, int r, int h,
                                               const mf_v8i (&Bf)[NTL][W / 2], const float (&cin)[16],
                                               float (&b)[NTL], float (&s)[NTL], uint32_t m88, uint32_t c22,
                                               uint32_t (&raw)[W / 2], const uint32_t*& nxt) {
  constexpr int KS = W / 2;
  // every tile starts from the SAME accumulator tuple (the MFMA reads it as its C operand: no copy).  The ragged last
  // tile does too and masks its missing rows behind the products -- a second tuple with -inf in those rows was kept in
  // 16 registers across the whole scan (hoisted out of the column-group loop) for the one tile that needs it.
  mf_v16f c0;
#pragma unroll
  for (int i = 0; i < 16; ++i) c0[i] = cin[i];
  mf_v8i Af[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) Af[k] = fp4_spread_from(raw[k], m88, c22);
  if (!TAIL) {
    // the next tile's rows: a running address, no clamp -- rows past Kf (the ragged last tile's missing ones, or a whole
    // tile that is never used when Kf is a multiple of 32) lie inside the workgroup's LDS block (at most 32 rows past
    // the staged descriptors: the counters behind them, sf_match_lds_bytes), spread to valid fp4 whatever their bits,
    // and are masked behind the products
    nxt += 32 * W;
    load_raw<KS>(nxt, raw);
  }
  // the scores kept so far move with the origin (the first row of the current tile).  Plain adds: rounds 2-4 paired them
  // in v_pk_add_f32 with the step in an SGPR pair -- packed f32 beside MFMAs costs more issue time than the two adds it
  // replaces (MI355X_MICROARCH.md, cycle constants), and since round 5 no packed-f32 instruction with a scalar source is
  // left in this translation unit (tools/pk_isa_scan.py --strict: the build gate of DESIGN.md section 3).
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    b[j] += 32.f * MF_FR;
    s[j] += 32.f * MF_FR;
  }
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    mf_v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[0], Bf[j][0], c0, 4, 4, 0, 0, 0, 0);
#pragma unroll
    for (int k = 1; k < KS; ++k)
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Af[k], Bf[j][k], acc, 4, 4, 0, 0, 0, 0);
    if (TAIL) {
      const int left = Kf - mt * 32 - 4 * h;      // rows of this lane half that exist: register i holds row (i & 3) + 8 (i >> 2)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = ((i & 3) + 8 * (i >> 2) < left) ? acc[i] : -INFINITY;
    }
    top2_update16(acc, b[j], s[j]);
  }
}

// kNN-2 of the "to" rows of NTL 32-column tiles over all "from" rows; on return lanes 0..31 hold, for
// column tile[j] * 32 + lane: d1 / d2 (Hamming, 0xFFFF when absent) and the from index of d1.
// The resident "to" operands of one scan: NTL column tiles spread to fp4, the columns' constants.
template <int W, int NTL>
struct MfB {
  mf_v8i Bf[NTL][W / 2];
  int tsum[NTL];
};

// Loads and spreads the "to" rows of the tiles tile[0..NTL) (a tile past the frame's rows is empty).
template <int W, int NTL>
__device__ __forceinline__ void mf_load_b(const uint32_t* __restrict__ dT, int Kt, const int (&tile)[NTL], int lane,
                                          MfB<W, NTL>& B) {
  constexpr int KS = W / 2;
  const int r = lane & 31, h = lane >> 5;
  uint32_t m88, c22;   // constants pinned in VGPRs (see fp4_spread)
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int t = tile[j] * 32 + r;
    uint32_t raw[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) raw[k] = 0;
    if (t < Kt) load_raw<KS>(dT + (size_t)t * W + KS * h, raw);
    int p = 0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      B.Bf[j][k] = fp4_spread_to(raw[k], m88, c22);
      p += __popc(raw[k] & 0x77777777u);
    }
    p += __shfl_xor(p, 32);                 // both halves of the row
    B.tsum[j] = 24 * W - 2 * p;             // the column's constant (see fp4_spread_from)
  }
}

// ---- round 5: the scan software-pipelined inside a wavefront (256-bit descriptors) -------------------------------------
// knn2_mfma_tile above runs, per column tile, 4 dependent MFMAs, waits for the result (s_nop 11), then 20 dependent
// vector instructions that consume it, on ONE accumulator tuple: nothing of a wavefront's own stream overlaps -- matrix-pipe
// time (4 x 32 cycles per 32 x 32 tile pair) and vector issue time (4 x 8 + 26 x 4; MI355X_MICROARCH.md, cycle constants)
// only overlap between the wavefronts that share a SIMD.
// Here the top-2 update of tile j - 1 is issued in the gaps of tile j's MFMAs, from a SECOND accumulator tuple: five to six
// vector instructions per 32-cycle gap (their issue cost 8 + 6 x 4 fits it), so a tile pair costs the matrix pipe's 128
// cycles plus what stays outside (the spread of the "from" tile, the loop).  Measured (profiles/r05s_*, r05zc_*): the
// cfg3-shaped launch 10.8 -> 10.0 ms per 100 000 pairs with the matrix pipe 75 % busy at the 1.69 GHz the chip sustains
// under it, and ONE workgroup per CU now reaches 96 % of the rate of three.  The MFMAs have to sit in the asm statements
// with the vector instructions (the compiler does not interleave an asm block with builtins), so the wait states are
// written out by hand (cdna_hip_programming.md section 5.7 item 2):
//   * a VALU-written A operand -> MFMA: s_nop 1 opens the first half (the spread is compiler code right in front); the
//     second half's A operands are inputs of the first half too, so they are written before it;
//   * an MFMA result -> a VALU reader: 12 states after the LAST MFMA of the tuple.  The old tuple's last MFMA is followed
//     by 6 vector instructions (the end of the second half), then s_nop 1 (2), the new tile's first MFMA (1), the two
//     origin shifts (2) and s_nop 1 (2) stand in front of the first read: 13.  The drain opens with s_nop 11;
//   * an accumulate chain (the MFMA takes the previous result whole as C) needs none.
// A statement's operands are limited to 30, hence two halves per tile (2 MFMAs + the update over 8 accumulator registers
// each).  Same operations in the same order on the same values as top2_update16: same bytes.
typedef int mf_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mf_pipe_half1(mf_v16f& acc, const mf_v4i& a0, const mf_v4i& a1, const mf_v4i& a2,
                                              const mf_v4i& a3, const mf_v4i& b0, const mf_v4i& b1, const mf_v16f& cin,
                                              const mf_v16f& old, float& b, float& s, float& x0, float& ta, float& tb) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 %[acc], %[a0], %[b0], %[cin] cbsz:4 blgp:4\n\t"
      "v_add_f32 %[b], 0x3c800000, %[b]\n\t"            // the origin moves by one tile: 32 / 2048
      "v_add_f32 %[s], 0x3c800000, %[s]\n\t"
      "s_nop 1\n\t"
      "v_med3_f32 %[x0], %[b], %[o0], %[o1]\n\t"
      "v_max3_f32 %[b], %[b], %[o0], %[o1]\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 %[acc], %[a1], %[b1], %[acc] cbsz:4 blgp:4\n\t"
      "v_med3_f32 %[ta], %[b], %[o2], %[o3]\n\t"
      "v_max3_f32 %[b], %[b], %[o2], %[o3]\n\t"
      "v_max3_f32 %[s], %[s], %[x0], %[ta]\n\t"
      "v_med3_f32 %[ta], %[b], %[o4], %[o5]\n\t"
      "v_max3_f32 %[b], %[b], %[o4], %[o5]\n\t"
      "v_med3_f32 %[tb], %[b], %[o6], %[o7]"
      : [acc] "=&v"(acc), [b] "+v"(b), [s] "+v"(s), [x0] "=&v"(x0), [ta] "=&v"(ta), [tb] "=&v"(tb)
      : [a0] "v"(a0), [a1] "v"(a1), "v"(a2), "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [cin] "v"(cin), [o0] "v"(old[0]),
        [o1] "v"(old[1]), [o2] "v"(old[2]), [o3] "v"(old[3]), [o4] "v"(old[4]), [o5] "v"(old[5]), [o6] "v"(old[6]),
        [o7] "v"(old[7]));
}
__device__ __forceinline__ void mf_pipe_half2(mf_v16f& acc, const mf_v4i& a2, const mf_v4i& a3, const mf_v4i& b2,
                                              const mf_v4i& b3, const mf_v16f& old, float& b, float& s, float& ta,
                                              float& tb) {
  float x0;
  asm volatile(
      "v_mfma_f32_32x32x64_f8f6f4 %[acc], %[a2], %[b2], %[acc] cbsz:4 blgp:4\n\t"
      "v_max3_f32 %[b], %[b], %[o6], %[o7]\n\t"
      "v_max3_f32 %[s], %[s], %[ta], %[tb]\n\t"
      "v_med3_f32 %[x0], %[b], %[o8], %[o9]\n\t"
      "v_max3_f32 %[b], %[b], %[o8], %[o9]\n\t"
      "v_med3_f32 %[ta], %[b], %[o10], %[o11]\n\t"
      "v_max3_f32 %[b], %[b], %[o10], %[o11]\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 %[acc], %[a3], %[b3], %[acc] cbsz:4 blgp:4\n\t"
      "v_max3_f32 %[s], %[s], %[x0], %[ta]\n\t"
      "v_med3_f32 %[ta], %[b], %[o12], %[o13]\n\t"
      "v_max3_f32 %[b], %[b], %[o12], %[o13]\n\t"
      "v_med3_f32 %[tb], %[b], %[o14], %[o15]\n\t"
      "v_max3_f32 %[b], %[b], %[o14], %[o15]\n\t"
      "v_max3_f32 %[s], %[s], %[ta], %[tb]"
      : [acc] "+v"(acc), [b] "+v"(b), [s] "+v"(s), [x0] "=&v"(x0), [ta] "+v"(ta), [tb] "+v"(tb)
      : [a2] "v"(a2), [a3] "v"(a3), [b2] "v"(b2), [b3] "v"(b3), [o6] "v"(old[6]), [o7] "v"(old[7]), [o8] "v"(old[8]),
        [o9] "v"(old[9]), [o10] "v"(old[10]), [o11] "v"(old[11]), [o12] "v"(old[12]), [o13] "v"(old[13]),
        [o14] "v"(old[14]), [o15] "v"(old[15]));
}
// the pending tuple's update with no tile behind it (end of the full tiles)
__device__ __forceinline__ void mf_pipe_drain(const mf_v16f& old, float& b, float& s) {
  float x0, ta, tb;
  asm volatile(
      "s_nop 11\n\t"
      "v_add_f32 %[b], 0x3c800000, %[b]\n\t"
      "v_add_f32 %[s], 0x3c800000, %[s]\n\t"
      "v_med3_f32 %[x0], %[b], %[o0], %[o1]\n\t"
      "v_max3_f32 %[b], %[b], %[o0], %[o1]\n\t"
      "v_med3_f32 %[ta], %[b], %[o2], %[o3]\n\t"
      "v_max3_f32 %[b], %[b], %[o2], %[o3]\n\t"
      "v_max3_f32 %[s], %[s], %[x0], %[ta]\n\t"
      "v_med3_f32 %[ta], %[b], %[o4], %[o5]\n\t"
      "v_max3_f32 %[b], %[b], %[o4], %[o5]\n\t"
      "v_med3_f32 %[tb], %[b], %[o6], %[o7]\n\t"
      "v_max3_f32 %[b], %[b], %[o6], %[o7]\n\t"
      "v_max3_f32 %[s], %[s], %[ta], %[tb]\n\t"
      "v_med3_f32 %[x0], %[b], %[o8], %[o9]\n\t"
      "v_max3_f32 %[b], %[b], %[o8], %[o9]\n\t"
      "v_med3_f32 %[ta], %[b], %[o10], %[o11]\n\t"
      "v_max3_f32 %[b], %[b], %[o10], %[o11]\n\t"
      "v_max3_f32 %[s], %[s], %[x0], %[ta]\n\t"
      "v_med3_f32 %[ta], %[b], %[o12], %[o13]\n\t"
      "v_max3_f32 %[b], %[b], %[o12], %[o13]\n\t"
      "v_med3_f32 %[tb], %[b], %[o14], %[o15]\n\t"
      "v_max3_f32 %[b], %[b], %[o14], %[o15]\n\t"
      "v_max3_f32 %[s], %[s], %[ta], %[tb]"
      : [b] "+v"(b), [s] "+v"(s), [x0] "=&v"(x0), [ta] "=&v"(ta), [tb] "=&v"(tb)
      : [o0] "v"(old[0]), [o1] "v"(old[1]), [o2] "v"(old[2]), [o3] "v"(old[3]), [o4] "v"(old[4]), [o5] "v"(old[5]),
        [o6] "v"(old[6]), [o7] "v"(old[7]), [o8] "v"(old[8]), [o9] "v"(old[9]), [o10] "v"(old[10]), [o11] "v"(old[11]),
        [o12] "v"(old[12]), [o13] "v"(old[13]), [o14] "v"(old[14]), [o15] "v"(old[15]));
}

// the full "from" tiles of a scan in the pipelined form (W = 8, NTL even: tile j writes tuple j & 1 and consumes the other)
template <int NTL>
__device__ __forceinline__ void mf_scan_full_tiles_pipe(const uint32_t* fromD, int n_full, const MfB<8, NTL>& B,
                                                        const float (&cin)[16], float (&b)[NTL], float (&s)[NTL],
                                                        uint32_t m88, uint32_t c22, uint32_t (&raw)[4],
                                                        const uint32_t*& nxt) {
  static_assert(NTL == 2 || NTL == 4, "tiles alternate between two accumulator tuples");
  if (n_full <= 0) return;
  mf_v4i B4[NTL][4];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) B4[j][k] = mf_v4i{B.Bf[j][k][0], B.Bf[j][k][1], B.Bf[j][k][2], B.Bf[j][k][3]};
  }
  mf_v16f c0, acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c0[i] = cin[i]; acc1[i] = -INFINITY; }     // nothing pending: an update that changes nothing
  for (int mt = 0; mt < n_full; ++mt) {
    mf_v4i A4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const mf_v8i a = fp4_spread_from(raw[k], m88, c22);
      A4[k] = mf_v4i{a[0], a[1], a[2], a[3]};
    }
    nxt += 32 * 8;                       // the next tile's rows (see knn2_mfma_tile)
    load_raw<4>(nxt, raw);
#pragma unroll
    for (int j = 0; j < NTL; j += 2) {
      constexpr int JP = NTL - 1;        // tile 0 consumes the LAST tile of the previous "from" tile
      float x0, ta, tb;
      const int jp = j == 0 ? JP : j - 1;
      mf_pipe_half1(acc0, A4[0], A4[1], A4[2], A4[3], B4[j][0], B4[j][1], c0, acc1, b[jp], s[jp], x0, ta, tb);
      mf_pipe_half2(acc0, A4[2], A4[3], B4[j][2], B4[j][3], acc1, b[jp], s[jp], ta, tb);
      mf_pipe_half1(acc1, A4[0], A4[1], A4[2], A4[3], B4[j + 1][0], B4[j + 1][1], c0, acc0, b[j], s[j], x0, ta, tb);
      mf_pipe_half2(acc1, A4[2], A4[3], B4[j + 1][2], B4[j + 1][3], acc0, b[j], s[j], ta, tb);
    }
  }
  mf_pipe_drain(acc1, b[NTL - 1], s[NTL - 1]);
}

// kNN-2 of the resident "to" columns over all "from" rows; on return lanes 0..31 hold, for column tile[j] * 32 + lane:
// d1 / d2 (Hamming, 0xFFFF when absent) and the from index of d1.
template <int W, int NTL, bool PIPE = false>
__device__ __forceinline__ void mf_scan(const uint32_t* fromD, int Kf, const MfB<W, NTL>& B, int lane,
                                        uint32_t (&d1)[NTL], uint32_t (&d2)[NTL], int (&idx)[NTL]) {
  constexpr int KS = W / 2;
  const int r = lane & 31, h = lane >> 5;
  uint32_t m88, c22;
  asm volatile("v_mov_b32 %0, 0x88888888" : "=v"(m88));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(c22));
  float cin[16], b[NTL], s[NTL];
#pragma unroll
  for (int i = 0; i < 16; ++i) cin[i] = -(float)((i & 3) + 8 * (i >> 2) + 4 * h) * MF_FR;
#pragma unroll
  for (int j = 0; j < NTL; ++j) { b[j] = -INFINITY; s[j] = -INFINITY; }
  const int n_full = Kf >> 5;
  uint32_t raw[KS];
  load_raw<KS>(fromD + (size_t)min(r, Kf - 1) * W + KS * h, raw);
  const uint32_t* nxt = fromD + (size_t)r * W + KS * h;
  if constexpr (PIPE && W == 8 && (NTL == 2 || NTL == 4)) {
    mf_scan_full_tiles_pipe<NTL>(fromD, n_full, B, cin, b, s, m88, c22, raw, nxt);
  } else {
    for (int mt = 0; mt < n_full; ++mt) knn2_mfma_tile<W, NTL, false>(fromD, Kf, mt, r, h, B.Bf, cin, b, s, m88, c22, raw, nxt);
  }
  if (Kf & 31) knn2_mfma_tile<W, NTL, true>(fromD, Kf, n_full, r, h, B.Bf, cin, b, s, m88, c22, raw, nxt);
  const float org = (float)(32 * (((Kf + 31) >> 5) - 1)) * MF_FR;
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const float ob = __shfl_xor(b[j], 32), os = __shfl_xor(s[j], 32);
    const float nb = fmaxf(b[j], ob) - org;
    const float ns = fmaxf(fminf(b[j], ob), fmaxf(s[j], os)) - org;
    const float dot1 = 2.f * ceilf(nb * 0.5f), dot2 = 2.f * ceilf(ns * 0.5f);
    idx[j] = (int)((dot1 - nb) * 2048.f);
    d1[j] = (uint32_t)((32 * W - B.tsum[j] - (int)dot1) >> 1);
    d2[j] = ns == -INFINITY ? 0xFFFFu : (uint32_t)((32 * W - B.tsum[j] - (int)dot2) >> 1);
  }
}

template <int W, int NTL, bool PIPE = false>
__device__ __forceinline__ void knn2_mfma(const uint32_t* fromD, int Kf, const uint32_t* __restrict__ dT, int Kt,
                                          const int (&tile)[NTL], int lane, uint32_t (&d1)[NTL], uint32_t (&d2)[NTL],
                                          int (&idx)[NTL]) {
  MfB<W, NTL> B;
  mf_load_b<W, NTL>(dT, Kt, tile, lane, B);
  mf_scan<W, NTL, PIPE>(fromD, Kf, B, lane, d1, d2, idx);
}

// Body of the matching stage for ONE pair (the calling workgroup); `smem` is the workgroup's dynamic
// LDS.  Returns whether the pair goes on to motion estimation (block-uniform).  With list == nullptr
// the pair is not appended to a work list (fused pipeline, k_verify.hip).
// `out` = the pair's correspondence list (kcap entries; global in the stage kernels, LDS -- it may alias the
// staged "from" block, which is dead by the time the list is written -- in the fused kernel); hdr_out / pass_out
// = the pair's header and pass-1 state (same two homes).
template <int W, int NQ, int NT, int MF_NTL = 2, bool MF_PIPE = false>
__device__ __forceinline__ bool match_v2_body(const StoreView& st, int pair, int sF, int sT, float nndr, int min_inliers,
                                              int est, uint32_t* out, CorrHeader& hdr_out, PassState& pass_out,
                                              int32_t* __restrict__ list, int32_t* __restrict__ counter, int* smem,
                                              unsigned long long* trace_row = nullptr) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  if ((unsigned)sF >= (unsigned)st.n_slots || (unsigned)sT >= (unsigned)st.n_slots) {
    if (tid == 0) {
      CorrHeader h = {0, 0, 0, 0};
      hdr_out = h;
      PassState ps;
#pragma unroll
      for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
      ps.var = 1.0; ps.var_ang = 1.0; ps.is_null = 1; ps.inliers = 0; ps.matches = 0; ps.pad = 0;
      pass_out = ps;
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

  // (Round 5 measured the staging by LDS-DMA with the first group's "to" rows loaded and spread beside it -- one memory
  //  latency in front of a pair's first MFMA instead of two, at 128 registers and no scratch: 22.6 against 22.8 M pairs/s,
  //  profiles/r05p_preload_ab.txt.  With four to five workgroups of other pairs on the CU a pair's own staging latency is
  //  already covered; not kept.)
  {
    const uint4* src = reinterpret_cast<const uint4*>(dF);
    uint4* dst = reinterpret_cast<uint4*>(fromD);
    for (int i = tid; i < Kf * (W / 4); i += NT) dst[i] = src[i];
  }
  for (int i = tid; i < Kf; i += NT) cnt[i] = 0;
  if (tid < 16) misc[tid] = 0;
  __syncthreads();
  SF_TRACE_ROW_MARK(trace_row, 32);   // "from" rows staged

  int rejected = 0;
  if constexpr (NQ == 0) {
    if (Kf > 0) {
      // "to" tiles resident per wavefront and scan: 2 keeps the fused kernel within 128 VGPRs (4 workgroups
      // per CU, which its motion-estimation chains need); the stage kernel takes 4 at 2 workgroups per CU
      // (the spread of the "from" tile is then shared by twice the columns: 0.293 -> 0.276 ms per 10 000 pairs)
      constexpr int NTL = W == 8 ? MF_NTL : 1;
      const int n_nt = (Kt + 31) >> 5;
      // a wavefront owns tiles wave, wave + NW, ...; it scans them in groups of up to NTL (a group that is
      // short of tiles drops to the next smaller instantiation; a 3-tile group runs as 4 with an empty tile)
      auto group = [&](auto g, int t0) {
        constexpr int G = decltype(g)::value;
        int tl[G], f[G];
        uint32_t a1[G], a2[G];
#pragma unroll
        for (int j = 0; j < G; ++j) tl[j] = t0 + j * NW;
        knn2_mfma<W, G, MF_PIPE>(fromD, Kf, dT, Kt, tl, lane, a1, a2, f);
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int t = tl[j] * 32 + lane;
          if (lane < 32 && t < Kt) {
            const bool acc = (Kf >= 2) && !((float)a1[j] > nndr * (float)a2[j]);
            if (acc) {
              atomicAdd(&cnt[f[j]], 1);
              owner[f[j]] = t;
            } else {
              ++rejected;
            }
          }
        }
      };
      for (int t0 = wave; t0 < n_nt; t0 += NW * NTL) {
        const int avail = (n_nt - t0 + NW - 1) / NW;
        if (NTL >= 4 && avail >= 3) group(std::integral_constant<int, 4>{}, t0);
        else if (NTL >= 2 && avail >= 2) group(std::integral_constant<int, 2>{}, t0);
        else group(std::integral_constant<int, 1>{}, t0);
      }
    }
  } else if (Kf > 0) {
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
  SF_TRACE_ROW_MARK(trace_row, 33);   // wavefront 0 done with its scans
  __syncthreads();
  SF_TRACE_ROW_MARK(trace_row, 34);   // all wavefronts done

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
  SF_TRACE_ROW_MARK(trace_row, 35);   // list compacted

  const int unique_to = (Kf > 0 && Kt > 0) ? misc[0] + n_corr : 0;
  const int words_from = (Kf > 0 && mF.y > 0) ? Kf : 0;
  const int words_to = (mT.y > 0) ? unique_to : 0;
  // est: 0 = 3D->3D gate (:1117-1118); 1 = PnP gate (:1070-1071, 2D words of the "to" frame);
  //      2 = PnP without a calibrated camera (:1059-1065): the estimation never runs
  //      3 = PnP, both directions (Vis/ForwardEstOnly = false): either gate; every pair with a correspondence goes on,
  //          the estimates count their own matches (the union of the two is not this kernel's to know)
  const bool motion = est == 0 ? (unique_to > 0 && words_from >= min_inliers && words_to >= min_inliers)
                    : est == 1 ? (unique_to > 0 && words_from >= min_inliers && unique_to >= min_inliers)
                               : (est == 3 && unique_to > 0 && ((words_from >= min_inliers && unique_to >= min_inliers) ||
                                                                (words_to >= min_inliers && Kf >= min_inliers)));
  const bool survivor = est == 3 ? (motion && n_corr > 0) : (motion && n_corr >= min_inliers && n_corr >= (est == 0 ? 3 : 4));
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
    hdr_out = h;
    PassState ps;
#pragma unroll
    for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
    ps.var = 1.0; ps.var_ang = 1.0;
    ps.is_null = 1;
    ps.inliers = 0;
    ps.matches = (motion && !survivor) ? misc[2] : 0;
    ps.pad = 0;
    pass_out = ps;
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
  const int pair = (int)blockIdx.x;
  match_v2_body<W, NQ, NT>(st, pair, pair_from[pair], pair_to[pair], nndr, min_inliers, est, corr + (size_t)pair * st.kcap,
                           hdr[pair], pass[pair], list, counter, smem);
}

// the matrix-core variant as a stage kernel: 4 column tiles per scan at 2 workgroups per CU (<= 256
// VGPRs; measured equal to 3 and 4 workgroups per CU -- the kernel is bound by VALU issue, not by
// occupancy).  A register budget <= 256 also keeps the MFMA results in VGPRs: with the full 512 the
// compiler accumulates in AGPRs and pays a v_accvgpr_read per value in the epilogue.
template <int W, int NT>
__global__ void __launch_bounds__(NT, 2)
k_match_global_mf(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
                  float nndr, int min_inliers, int est, uint32_t* __restrict__ corr, CorrHeader* __restrict__ hdr,
                  PassState* __restrict__ pass, int32_t* __restrict__ list, int32_t* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  const int pair = (int)blockIdx.x;
  match_v2_body<W, 0, NT, 4>(st, pair, pair_from[pair], pair_to[pair], nndr, min_inliers, est,
                             corr + (size_t)pair * st.kcap, hdr[pair], pass[pair], list, counter, smem);
}

template <int W, int NQ, int NT>
void launch_match_v2(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  size_t lds = sf_match_lds_bytes(st.kcap, W);
  if (const char* v = getenv("SF_MATCH_LDS_PAD")) lds += (size_t)atoi(v);   // occupancy experiment (diagnostic)
  int32_t* counters = (int32_t*)c->counters.p;
  if constexpr (NQ == 0)
    hipLaunchKernelGGL((k_match_global_mf<W, NT>), dim3(n), dim3(NT), lds, c->stream, st, d_from, d_to,
                       c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p,
                       (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
  else
    hipLaunchKernelGGL((k_match_global_v2<W, NQ, NT>), dim3(n), dim3(NT), lds, c->stream, st, d_from, d_to,
                       c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p,
                       (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
}

template <int W, int NQ, int NT>
void launch_match(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  const size_t lds = (size_t)(2 * st.kcap + 16) * sizeof(int);
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_match_global<W, NQ, NT>), dim3(n), dim3(NT), lds, c->stream, st, d_from, d_to,
                     c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p,
                     (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
}

template <int W>
void launch_match_l2(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  const size_t lds = (size_t)(2 * st.kcap + 16) * sizeof(int);
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_match_global<W, 1, 256, true>), dim3(n), dim3(256), lds, c->stream, st, d_from, d_to,
                     c->dparams.nndr, c->dparams.min_inliers, sf_est_mode(c), (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p,
                     (PassState*)c->pass1.p, (int32_t*)c->list1.p, counters + 0);
}

}  // namespace

int sf_launch_match_global(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  if (n <= 0) return SF_OK;
  if (c->params.desc_type == 1) {       // float32 rows: exact L2 on the VALU (north_star: MFMA only for the NetVLAD matrix)
    sf_prof_begin(c, SF_K_MATCH);
    if (st.w == 64) launch_match_l2<64>(c, st, d_from, d_to, n);
    else launch_match_l2<128>(c, st, d_from, d_to, n);
    sf_prof_end(c, SF_K_MATCH);
    SF_HIP(c, hipGetLastError());
    return SF_OK;
  }
  // geometry: variant = NQ * 1000 + NT (tunable through SF_MATCH_VARIANT for A/B runs)
  int variant = c->match_variant;
  if (variant == 0) {
    // default: LDS + u16 variant while the staged "from" block keeps >= 2 workgroups per CU
    const size_t lds_v2 = sf_match_lds_bytes(st.kcap, st.w);
    variant = lds_v2 <= 64 * 1024 ? (c->match_mfma && st.kcap <= MF_MAX_ROWS ? 10256 : 12256) : 2256;
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
    SF_CASE2(0, 256)   // NQ = 0: fp4 matrix-core variant
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
