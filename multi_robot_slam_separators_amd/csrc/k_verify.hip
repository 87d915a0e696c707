// k_verify.hip -- the verification kernels as ONE translation unit, plus the fused per-pair pipeline.
//
// The stage files are included (not linked) so that their per-pair bodies can be inlined into
// k_verify_fused: one 256-thread workgroup takes a candidate pair through the whole of
// StereoCamGeometricTools::estimateTransformation (stereoCamGeometricTools.cpp:122-178) --
// global matching, RANSAC, guess-guided matching, RANSAC again, result assembly -- without leaving the
// CU.  Why fuse: the motion-estimation stages are short dependent fp64 chains that occupy a few
// hundred workgroups for tens of microseconds each and leave the issue ports idle, while matching is bound by
// instruction issue (matrix pipe + VALU top-2 scan).  In the fused kernel the 20 % of pairs that survive
// matching run their chain while the other workgroups of the same CU are still matching.
//
// Round 2: everything a pair's stages hand to each other -- the correspondence lists, their headers, the pass
// states, the guided flag -- stays in the workgroup's LDS; the only global write of a pair is its sf_result
// (368 B).  Round 1 round-tripped all of it through HBM between stages of the SAME workgroup (a write, a
// vmcnt(0) wait, a barrier and a dependent read per hand-over, 60 MB of writes per 10 000-pair launch).
// SF_OPT_DEBUG_CORR (sf_debug_correspondences, tests) additionally copies lists and headers to the global
// workspace.  The stage kernels remain for the shapes the fused / split forms do not take and as the A/B
// reference (SF_FUSED=0): all paths run the same bodies and produce identical bytes.
//
// Round 5: the bundle adjustment (myRegistrationVis.cpp:1192-1370) is a launch of its own (k_ba_pass) in every
// pipeline.  Inside the estimators' kernels it had forced a second instantiation of each of them -- 256 registers +
// 700 B of scratch per lane at two workgroups per CU -- and the reference's as-shipped flow (PnP + adjustment) ran
// at 3.1 M pairs/s against 13.7 M without it.  With the adjustment on, a survivor's chain is cut in two:
// [estimate 1] -> k_ba_pass -> [guided matching + estimate 2] -> k_ba_pass (+ result), the estimates leaving their
// inlier sets as one byte per "from" feature.
// Compiled with -ffp-contract=off (canonical arithmetic of the RANSAC / guided bodies).
// Workgroups of k_chain_pnp per CU the compiler budgets registers for.  Measured on the bench's PnP line (round 4,
// profiles/r04z_pnp_chain_occupancy.txt): 2 (256 registers, no scratch) 11.8 M pairs/s, 3 (168 registers, 116 B of
// scratch per lane) 13.7 M, 4 (128 registers, 296 B) 13.4 M -- the chains are latency-bound, more of them side by side
// is worth more than their spills cost, until the matching launch beside them loses the slots.
#ifndef SF_PNP_CHAIN_OCC
#define SF_PNP_CHAIN_OCC 3
#endif
#ifndef SF_MATCH_PIPE
#define SF_MATCH_PIPE 1          // k_match_split: the scan software-pipelined inside a wavefront (k_match.hip, mf_pipe_*)
#endif
#ifndef SF_PNP_CHAIN_OCC4
#define SF_PNP_CHAIN_OCC4 2      // the four-wavefront form's budget (3 = the build that failed a parity test: see k_chain_pnp)
#endif
#include "k_match.hip"
#include "k_ransac.hip"
#include "k_guided.hip"
#include "k_pnp.hip"

namespace {

// Tail of the fused kernel's dynamic LDS: what the stages of one pair hand to each other.
struct FusedTail {
  PassState pass1, pass2;
  CorrHeader hdr1, hdr2;
  uint8_t guided_flag;
  int32_t stream_slot;       // accepted-result stream: this pair's slot in the host block (-1: not accepted / full)
};

// What the two halves of a chain cut around the bundle adjustment leave in HBM besides lists, headers and pass states
struct BaHandover {
  uint8_t* mask1;     // [pairs][kcap] inlier bytes of the first estimate, per "from" feature (zeroed before the launch)
  uint8_t* mask2;     // ... of the second
  uint8_t* est2;      // [pairs] 1 = the pair's second estimate ran (its state is to be adjusted)
};

// the accepted result of a finished pair leaves for the host NOW (posted PCIe writes beside the other pairs' work)
// instead of through a compaction kernel behind the launch: thread 0 takes the slot, 23 lanes move the 368 bytes
__device__ __forceinline__ void stream_accepted(int pair, const sf_result* __restrict__ out, const DeviceParams& P,
                                                int32_t& s_slot) {
  const AcceptStream& S = P.accept;
  const int tid = (int)threadIdx.x;
  if (tid == 0) {
    const bool ok = out[pair].success != 0;
    if (S.flags) S.flags[pair] = ok ? 1 : 0;
    int slot = -1;
    if (ok) {
      const unsigned sl = atomicAdd(S.counter, 1u);
      if (sl < (unsigned)S.cap) { slot = (int)sl; S.index[sl] = pair; }
    }
    s_slot = slot;
  }
  __syncthreads();
  const int slot = s_slot;
  if (slot >= 0 && tid < (int)(sizeof(sf_result) / 16)) {
    const uint4 v = reinterpret_cast<const uint4*>(out + pair)[tid];
    reinterpret_cast<uint4*>(S.records + slot)[tid] = v;
    if (S.records2) reinterpret_cast<uint4*>(S.records2 + slot)[tid] = v;
  }
}

// Everything after the pass-1 correspondence list of a pair: RANSAC, guess-guided matching, RANSAC, result.
// PART 0: the whole chain.  With the bundle adjustment on the chain is cut around its two launches (k_ba_pass):
// PART 1 = the first estimate (its state and inlier bytes go to HBM), PART 2 = guided matching + the second estimate
// from the ADJUSTED first state (list, header, flag, state, inlier bytes to HBM; the result is assembled by the second
// adjustment's launch).
template <int W, int NW = 4, int PART = 0>
__device__ __forceinline__ void chain_after_match(const StoreView& st, int pair, int sF, int sT, bool est1, FusedTail& T,
                                                  uint32_t* cl, unsigned char* chain_lds, uint32_t* __restrict__ corr2,
                                                  CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass1,
                                                  PassState* __restrict__ pass2, uint8_t* __restrict__ guided_flag,
                                                  sf_result* __restrict__ out, const DeviceParams& P,
                                                  const BaHandover& H) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int kcap = st.kcap;
  // from here on this workgroup is a short chain of dependent fp64 steps: let its wavefronts win the
  // issue arbitration against the matching wavefronts it shares SIMDs with (they are throughput-bound
  // and lose nothing measurable), so the chain -- the tail of the launch -- finishes sooner
  __builtin_amdgcn_s_setprio(3);
  if constexpr (PART != 2) {
    if (est1) {
      ransac_body<0, NW>(st, pair, sF, sT, cl, T.hdr1.n_corr, T.pass1, P, chain_lds, 2,
                         PART == 1 ? H.mask1 + (size_t)pair * kcap : nullptr);
      if constexpr (PART == 0) {
        if (P.force_3dof && tid == 0) pass_to3dof(T.pass1, 2);    // myRegistration.cpp:269-276, then :245-248 as pass 2's guess
      }
      __syncthreads();
    }
    if constexpr (PART == 1) {
      if (tid == 0) pass1[pair] = T.pass1;        // (Reg/Force3DoF's applications: behind the adjustment, k_ba_pass)
      return;
    }
  }
  // pass 2: guess-guided matching (:476-825) seeded with the pass-1 pose, RANSAC again
  const bool est2 = guided_body<W, false, NW>(st, pair, sF, sT, T.pass1, T.pass2, T.guided_flag, cl, T.hdr2, nullptr, nullptr,
                                              P, reinterpret_cast<int*>(chain_lds));
  __syncthreads();
  if (P.dbg_corr || PART == 2) {
    const int n = T.hdr2.n_corr;
    for (int i = tid; i < n; i += NT) corr2[(size_t)pair * kcap + i] = cl[i];
    if (tid == 0) hdr2[pair] = T.hdr2;
  }
  if (est2) {
    ransac_body<0, NW>(st, pair, sF, sT, cl, T.hdr2.n_corr, T.pass2, P, chain_lds, 11,
                       PART == 2 ? H.mask2 + (size_t)pair * kcap : nullptr);
    if constexpr (PART == 0) {
      if (P.force_3dof && tid == 0) pass_to3dof(T.pass2, 1);
    }
    __syncthreads();
  }
  SF_TRACE_MARK(P, pair, 17);
  if constexpr (PART == 2) {
    if (tid == 0) { pass2[pair] = T.pass2; guided_flag[pair] = T.guided_flag; H.est2[pair] = est2 ? 1 : 0; }
    return;
  }
  if (tid == 0) {
    if (P.dbg_corr) { pass1[pair] = T.pass1; pass2[pair] = T.pass2; guided_flag[pair] = T.guided_flag; }
    finalize_one(T.pass1, T.pass2, T.guided_flag, out[pair]);
  }
  if (P.accept_on) stream_accepted(pair, out, P, T.stream_slot);
}

// WIDE: the instantiation for frames whose LDS working set does not let four workgroups share a CU anyway (K = 1000
// features: 52 KB, three per CU): compiled for two workgroups per CU it takes 162 registers -- three still fit -- and
// keeps FOUR resident "to" tiles per wavefront in the scan (one spread of a "from" tile per four tiles instead of two).
template <int W, int NQ, bool WIDE = false>
__global__ void __launch_bounds__(SF_BLOCK, WIDE ? 2 : 4)
k_verify_fused(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
               uint32_t* __restrict__ corr1, CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
               uint32_t* __restrict__ corr2, CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2,
               uint8_t* __restrict__ guided_flag, sf_result* __restrict__ out, DeviceParams P, int tail_off,
               PairSource src) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int kcap = st.kcap;
  int sF, sT;
  if (src.cand) {               // (what k_spec_pairs would have written to pair_from / pair_to)
    sF = -1; sT = -1;
    if ((unsigned)pair < *src.count) {
      const uint2 rc = src.cand[pair];
      if ((int)rc.x < src.n_l && (int)rc.y < src.n_r) {
        const int f = src.slot_other + (int)rc.y, t = src.slot_local + (int)rc.x;
        if ((unsigned)f < (unsigned)src.n_slots && (unsigned)t < (unsigned)src.n_slots) { sF = f; sT = t; }
      }
    }
  } else {
    sF = pair_from[pair]; sT = pair_to[pair];
  }
  FusedTail& T = *reinterpret_cast<FusedTail*>(smem_raw + tail_off);
  uint32_t* cl = reinterpret_cast<uint32_t*>(smem_raw);          // [kcap] the current pass's correspondence list
  unsigned char* chain_lds = smem_raw + (size_t)kcap * 4;        // RANSAC / guided working set behind it
  SF_TRACE_MARK(P, pair, 0);
  // pass 1: global matching (myRegistrationVis.cpp:826-895) and, for survivors, RANSAC (:1113-1152)
  const bool est1 = match_v2_body<W, NQ, SF_BLOCK, WIDE ? 4 : 2>(st, pair, sF, sT, P.nndr, P.min_inliers, 0, cl, T.hdr1, T.pass1,
                                                   nullptr, nullptr, reinterpret_cast<int*>(smem_raw),
                                                   P.dbg_trace ? P.dbg_trace + (size_t)pair * SF_TRACE_SLOTS : nullptr);
  __syncthreads();   // list / header / pass-1 defaults visible to the whole workgroup
  SF_TRACE_MARK(P, pair, 1);
  if (P.dbg_corr) {
    const int n = T.hdr1.n_corr;
    for (int i = tid; i < n; i += SF_BLOCK) corr1[(size_t)pair * kcap + i] = cl[i];
    if (tid == 0) hdr1[pair] = T.hdr1;
  }
  chain_after_match<W>(st, pair, sF, sT, est1, T, cl, chain_lds, corr2, hdr2, pass1, pass2, guided_flag, out, P,
                       BaHandover{nullptr, nullptr, nullptr});
}

// ---- the split pipeline: ONE matching launch over all pairs, ONE chain launch over the survivors -------------
// The fused kernel holds a latency-bound chain (~55 us alone, ~73 us beside matching workgroups) in a quarter of a
// CU's slots while the matching of the other pairs queues behind it, and the chains born from the last matches run on
// an emptying chip.  Split, the matching of all pairs runs with 4 "to" tiles resident per wavefront at three workgroups
// per CU (160 VGPRs: the spread of a "from" tile is shared by twice the columns of the fused kernel's scan), the
// survivors' lists / headers take one trip through HBM (~2 KB per survivor) and their chains run four to a CU.  Same
// bodies, same bytes.  On ONE stream the fused kernel is the faster form (its chains overlap other pairs' matching
// inside the launch); since sf_step_issue alternates the steps between two streams the neighbouring step fills a
// launch's tail anyway and the split form wins -- sf_use_split (sf_api.hip) picks it there (SF_OPT_STEP_SPLIT).
template <int W, int NTL = 4, int MINW = 3>
__global__ void __launch_bounds__(SF_BLOCK, MINW)
k_match_split(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
              uint32_t* __restrict__ corr1, CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
              CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2, uint8_t* __restrict__ guided_flag,
              int32_t* __restrict__ list, int32_t* __restrict__ counter, sf_result* __restrict__ out, DeviceParams P,
              int est) {
  extern __shared__ __attribute__((aligned(16))) int smem_i[];
  const int pair = blockIdx.x;
  SF_TRACE_MARK(P, pair, 0);
  const bool survivor = match_v2_body<W, 0, SF_BLOCK, NTL, SF_MATCH_PIPE != 0>(st, pair, pair_from[pair], pair_to[pair], P.nndr, P.min_inliers,
                                                         est, corr1 + (size_t)pair * st.kcap, hdr1[pair], pass1[pair],
                                                         list, counter, smem_i,
                                                         P.dbg_trace ? P.dbg_trace + (size_t)pair * SF_TRACE_SLOTS : nullptr);
  SF_TRACE_MARK(P, pair, 36);
  if (!survivor && threadIdx.x == 0) {
    // no motion estimate: guided matching is not eligible (its pass state is pass 1's), the result is final
    const PassState p = pass1[pair];
    if (P.dbg_corr) {
      const CorrHeader h = {0, 0, 0, 0};
      hdr2[pair] = h; pass2[pair] = p; guided_flag[pair] = 0;
    }
    finalize_one(p, p, 0, out[pair]);
    if (P.accept_on && P.accept.flags) P.accept.flags[pair] = 0;      // (accepted-result stream: not accepted)
  }
}

// (Round 5 also built a PERSISTENT form of this launch -- resident workgroups walking the pairs, the next pair's "from" rows
//  arriving in a second LDS buffer by LDS-DMA during the scan -- and short-lived workgroups of 2 / 4 pairs on it.  Correct,
//  slower in every shape measured (21.4 / 20.9 against 22.7 M pairs/s: profiles/r05m_*, r05y_*; docs/notebook_r05.md
//  sections 5 and 7): the second row buffer and 100-150 B of scratch cost more than the staging they hide.  Removed; the
//  last commit that holds it is "Matcher experiments: k pairs per workgroup ...".)

// NW: wavefronts per chain.  4 = the round-2 form (one 256-thread workgroup per survivor, 128 registers: a chain holds
// a quarter of a CU's wave slots and registers while three of its four wavefronts wait at barriers most of the time).
// 1 / 2 (round 5): one or two WAVEFRONTS per survivor -- with one, no barrier is left in the chain (a single-wavefront
// workgroup's __syncthreads is a wait on its own LDS traffic) and no wavefront waits for the one that solves.  The sums
// keep the canonical 256-lane order (sfd::canon_reduce plays the four wavefronts one after the other), so the results
// are the same bytes (tests/test_gpu_verify.py::test_chain_widths_give_the_same_bytes).  Measured on the bench step
// (profiles/r05a_chain_width.txt): a chain takes ~1.8 x as long on one wavefront as on four, and what caps the chains
// per CU is their LDS (32 KB each: five per CU whatever their width), not registers or wave slots -- 20.1 / 22.0 / 22.3 M
// pairs/s for 1 / 2 / 4.  The narrow forms stay selectable (SF_CHAIN_NW) for shapes whose working set is small.
template <int W, int NW, int PART>
__global__ void __launch_bounds__(64 * NW, 4)
k_chain(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
        const uint32_t* __restrict__ corr1, const CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
        uint32_t* __restrict__ corr2, CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2,
        uint8_t* __restrict__ guided_flag, const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
        sf_result* __restrict__ out, DeviceParams P, int tail_off, BaHandover H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x >= *counter) return;           // (the grid is sized for every pair surviving)
  const int pair = list[blockIdx.x];
  const int tid = threadIdx.x;
  const int kcap = st.kcap;
  const int sF = pair_from[pair], sT = pair_to[pair];
  FusedTail& T = *reinterpret_cast<FusedTail*>(smem_raw + tail_off);
  uint32_t* cl = reinterpret_cast<uint32_t*>(smem_raw);
  unsigned char* chain_lds = smem_raw + (size_t)kcap * 4;
  if constexpr (PART != 2) {
    const CorrHeader h1 = hdr1[pair];
    if (tid == 0) { T.hdr1 = h1; T.pass1 = pass1[pair]; }
    for (int i = tid; i < h1.n_corr; i += 64 * NW) cl[i] = corr1[(size_t)pair * kcap + i];
  } else {
    if (tid == 0) T.pass1 = pass1[pair];              // (the adjusted first estimate)
  }
  __syncthreads();
  SF_TRACE_MARK(P, pair, 1);
  chain_after_match<W, NW, PART>(st, pair, sF, sT, true, T, cl, chain_lds, corr2, hdr2, pass1, pass2, guided_flag, out, P, H);
}

template <int W, int NQ, bool WIDE = false>
int launch_fused(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                 size_t lds, int tail_off) {
  bool& attr_set = c->fused_attr[W == 16][NQ == 0][WIDE];   // one flag per instantiation
  if (lds > 64 * 1024 && !attr_set) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_verify_fused<W, NQ, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_verify_fused<W, NQ, WIDE>), dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                     (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (uint32_t*)c->corr2.p,
                     (CorrHeader*)c->hdr2.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p, d_out, c->dparams,
                     tail_off, c->pair_src);
  return SF_OK;
}

// offset of the FusedTail = the largest stage's working set (the chain stages sit behind the kcap-entry list)
size_t fused_tail_offset(const sf_context* c, const StoreView& st, bool with_match = true) {
  const int nc = c->dparams.grid_gx * c->dparams.grid_gy;
  const size_t match = with_match ? sf_match_lds_bytes(st.kcap, st.w) : 0;
  const size_t guided = (size_t)st.kcap * 4 + sf_guided_lds_bytes(st.kcap, nc);
  const size_t ransac = (size_t)st.kcap * 4 + ((sf_ransac_lds_bytes(st.kcap, c->dparams.iterations) + 15) & ~(size_t)15);
  return (std::max(match, std::max(guided, ransac)) + 15) & ~(size_t)15;
}

}  // namespace

// Dynamic LDS of the fused kernel = the largest stage + the hand-over tail; 0 when the fused pipeline does not
// apply (PnP estimator, SF_FUSED=0, or a stage that needs more than the 160 KB of a CU).
size_t sf_fused_lds_bytes(const sf_context* c, const StoreView& st) {
  if (!c->fused || c->dparams.estimation_type != 0 || c->dparams.bidirectional) return 0;   // (both directions: stage kernels)
  if (c->params.desc_type != 0) return 0;     // float32 descriptors: the stage kernels (exact L2 on the VALU)
  const size_t match = sf_match_lds_bytes(st.kcap, st.w);
  // same rule as sf_launch_match_global: the LDS-staged matching body only while the staged "from"
  // block leaves room for >= 2 workgroups per CU; beyond that the stage kernels (scalar-load matcher)
  if (match > 64 * 1024 || c->match_variant != 0) return 0;
  const size_t lds = fused_tail_offset(c, st) + ((sizeof(FusedTail) + 15) & ~(size_t)15);
  return lds <= 160 * 1024 ? lds : 0;
}

// (with the bundle adjustment on the fused kernel does not apply: the chain is cut around the adjustment's launches,
//  sf_launch_verify_split, or the stage kernels run)
int sf_launch_verify_fused(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n,
                           sf_result* d_out) {
  if (n <= 0) return SF_OK;
  const size_t lds = sf_fused_lds_bytes(c, st);
  if (lds == 0 || c->dparams.bundle_adjustment) return sf_fail(c, SF_EINVAL, "fused verification pipeline not applicable");
  const int tail_off = (int)fused_tail_offset(c, st);
  int rc;
  sf_prof_begin(c, SF_K_FUSED);
  const bool mf = c->match_mfma && st.kcap <= MF_MAX_ROWS;
  // more than a quarter of a CU's LDS per workgroup: four do not fit, so the build for fewer workgroups (162 registers:
  // up to three per CU) with four resident tiles
  static const bool wide_off = getenv("SF_FUSED_WIDE_OFF") != nullptr;      // (A/B runs)
  static const bool wide_all = getenv("SF_FUSED_WIDE_ALL") != nullptr;    // (experiment: the wide build for every shape)
  if (mf && st.w == 8 && (lds * 4 > 160 * 1024 || wide_all) && !wide_off)
    rc = launch_fused<8, 0, true>(c, st, d_from, d_to, n, d_out, lds, tail_off);
  else if (mf) rc = st.w == 8 ? launch_fused<8, 0>(c, st, d_from, d_to, n, d_out, lds, tail_off)
                              : launch_fused<16, 0>(c, st, d_from, d_to, n, d_out, lds, tail_off);
  else rc = st.w == 8 ? launch_fused<8, 2>(c, st, d_from, d_to, n, d_out, lds, tail_off)
                      : launch_fused<16, 2>(c, st, d_from, d_to, n, d_out, lds, tail_off);
  sf_prof_end(c, SF_K_FUSED);
  if (rc != SF_OK) return rc;
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

namespace {

// The PnP estimator's chain for the survivors of k_match_split: PnP -> guess-guided matching -> PnP -> result in ONE
// launch (the stage path: k_pnp, k_guided over ALL pairs, k_pnp, k_finalize).  The bodies keep handing their lists
// and pass states over through the global workspace (k_pnp's 168 registers leave no room for the LDS-resident form of
// the 3D-3D chain); what the fusion saves is three launches, the guided kernel's 10 000 workgroups that find nothing
// to do, and the gaps between them.  Same bodies, same bytes.  PART as in chain_after_match (bundle adjustment on:
// 1 = the first estimate, 2 = guided matching + the second estimate; the adjustments' launches in between and behind).
// NW: wavefronts per chain (SF_CHAIN_NW).  The narrow forms recompute the three bearings a P3P hypothesis needs instead
// of keeping one per correspondence in LDS (28 KB instead of 40 KB at K = 500: five chains per CU instead of four by LDS,
// six by registers), and no wavefront waits while another one solves.
// Register budget: the two-wavefront form (the default, 14.5 against 13.4 M pairs/s on the bench's PnP line,
// profiles/r05i_pnp_chain_width.txt) is compiled for SF_PNP_CHAIN_OCC = 3 wavefronts per SIMD (168 registers, ~120 B of
// scratch per lane) like round 4's chain.  The FOUR-wavefront form is compiled for 2 (256 registers, no scratch): at 3
// this round's build of it -- same bodies as the stage kernel k_pnp, which is correct at that budget -- returned wrong
// inlier sets from rtabmap's refinement rounds (tests/test_gpu_pnp.py::test_pnp_refinement_rounds with
// SF_CHAIN_PNP_NW=4; 7 inliers against the oracle's 35), and correct ones at 2.  profiles/r05r_pnp_chain_occ3.txt: the
// same source at 3 is correct again with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 and with -mllvm -no-stack-slot-sharing, so
// the fault belongs to ONE register allocation (106 SGPRs, 156 v_writelane / 427 v_readlane spill moves beside 128 B of
// scratch), not to the source; lane sharing, hazard wait states, foreign writers of the spill registers and lane moves
// under EXEC = 0 were checked on that build's ISA and are not it (tools/spill_lane_check.py, tools/ubench/lane_exec0.hip).
// Which instruction is wrong is not established; -DSF_PNP_CHAIN_OCC4=3 rebuilds the failing form, which is not shipped,
// and the parity soak (tools/soak_parity.py pnp) runs on the shipped widths.
template <int W, int PART, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? SF_PNP_CHAIN_OCC4 : SF_PNP_CHAIN_OCC)
k_chain_pnp(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
            const uint32_t* __restrict__ corr1, const CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
            uint32_t* __restrict__ corr2, CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2,
            uint8_t* __restrict__ guided_flag, const int32_t* __restrict__ list, const int32_t* __restrict__ counter,
            sf_result* __restrict__ out, DeviceParams P, BaHandover H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x >= *counter) return;
  const int pair = list[blockIdx.x];
  const int sF = pair_from[pair], sT = pair_to[pair];
  const int kcap = st.kcap;
  __builtin_amdgcn_s_setprio(3);
  SF_TRACE_MARK(P, pair, 1);
  if constexpr (PART != 2) {
    pnp_body<0, NW>(st, pair, pair_from, pair_to, corr1, hdr1, pass1, P, smem_raw, 37,
                    PART == 1 ? H.mask1 + (size_t)pair * kcap : nullptr);
    if constexpr (PART == 1) return;       // (Reg/Force3DoF's applications: behind the adjustment, k_ba_pass)
    if (P.force_3dof && threadIdx.x == 0) pass_to3dof(pass1[pair], 2);
    __syncthreads();     // pass1[pair] (written by thread 0) is read by every lane below
  }
  const bool est2 = guided_body<W, false, NW>(st, pair, sF, sT, pass1[pair], pass2[pair], guided_flag[pair],
                                              corr2 + (size_t)pair * kcap, hdr2[pair], nullptr, nullptr, P,
                                              reinterpret_cast<int*>(smem_raw));
  __syncthreads();
  SF_TRACE_MARK(P, pair, 42);
  if (est2) {
    pnp_body<0, NW>(st, pair, pair_from, pair_to, corr2, hdr2, pass2, P, smem_raw, 43,
                    PART == 2 ? H.mask2 + (size_t)pair * kcap : nullptr, true);
    if constexpr (PART == 0) {
      if (P.force_3dof && threadIdx.x == 0) pass_to3dof(pass2[pair], 1);
    }
    __syncthreads();
  }
  SF_TRACE_MARK(P, pair, 17);
  if constexpr (PART == 2) {
    if (threadIdx.x == 0) H.est2[pair] = est2 ? 1 : 0;
    return;
  }
  if (threadIdx.x == 0) finalize_one(pass1[pair], pass2[pair], guided_flag[pair], out[pair]);
  if (P.accept_on) {              // accepted-result stream (see chain_after_match)
    __syncthreads();              // (the bodies are done with the LDS: its first word carries the slot)
    stream_accepted(pair, out, P, *reinterpret_cast<int32_t*>(smem_raw));
  }
}

// ---- bundle adjustment as a launch of its own (k_ba.hip: ba_pass_body) ----------------------------------------------------
// One workgroup of NW wavefronts per entry of the pass's work list.  `run` (may be null): one byte per pair, 0 = this
// pair's pass has no estimate to adjust (pass 2 of a pair whose guided matching did not go on to the estimation).
// fin: the adjustment of pass 2 also assembles the pair's result (finalize_one) and, where the accepted-result stream is
// armed, hands it to the host -- the tail of the chain kernels.
// Two launches per pass where a keyframe holds more than SF_BA_SMALL_CAP features: the SMALL one takes every pair
// whose estimate has at most that many inliers (the adjustment's words) with an LDS working set sized for them -- 20 KB
// instead of 38 KB at K = 500: eight workgroups per CU instead of four, and one word per lane of the canonical 256-lane
// sums, so no accumulator stays live across a loop -- the other launch the pairs with more (none on typical frames: its
// workgroups read a pass state and leave).  A pair is handled, and its result assembled, by exactly one of the two.
#define SF_BA_SMALL_CAP 256
template <int NW, bool PNP, bool SMALL, int OCC = 1>
__global__ void __launch_bounds__(64 * NW, OCC)
k_ba_pass(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
          const int32_t* __restrict__ list, const int32_t* __restrict__ counter, const uint32_t* __restrict__ corr,
          const CorrHeader* __restrict__ hdr, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ run,
          PassState* __restrict__ pass, int extra_3dof, int fin, const PassState* __restrict__ pass1,
          const uint8_t* __restrict__ guided_flag, sf_result* __restrict__ out, DeviceParams P, int cap, int both) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((int)blockIdx.x >= *counter) return;
  const int pair = list[blockIdx.x];
  const int kcap = st.kcap;
  const PassState p0 = pass[pair];
  // :1192-1197 gate (the words3From / wordsTo conditions hold whenever the estimate ran); block-uniform
  const bool adjust = (!run || run[pair]) && !p0.is_null && p0.inliers > 0;
  const bool big = adjust && p0.inliers > SF_BA_SMALL_CAP;
  if (both && big == SMALL) return;            // the other launch's pair
  if (adjust) {
    if (p0.inliers <= cap)
      ba_pass_body<NW, PNP, SMALL>(st, pair_from[pair], pair_to[pair], corr + (size_t)pair * kcap, hdr[pair].n_corr,
                                   mask + (size_t)pair * kcap, p0, pass[pair], P, smem_raw, cap);
  }
  if (!run || run[pair]) {
    // myRegistration.cpp:269-276, and for pass 1 the application its result meets as the guess of pass 2 (:245-248)
    // (thread 0 wrote the adjusted state: same thread, program order)
    if (extra_3dof && threadIdx.x == 0) pass_to3dof(pass[pair], extra_3dof);
  }
  if (!fin) return;
  __syncthreads();
  if (threadIdx.x == 0) finalize_one(pass1[pair], pass[pair], guided_flag[pair], out[pair]);
  if (P.accept_on) {
    __syncthreads();              // (the adjustment is done with the LDS: its first word carries the slot)
    stream_accepted(pair, out, P, *reinterpret_cast<int32_t*>(smem_raw));
  }
}

}  // namespace

// The adjustment of pass `pass` (1 / 2) of the n pairs of a launch sequence, over that pass's work list.  mask: the
// estimates' inlier bytes [n][kcap]; run: see k_ba_pass; list_sel: which work list (1: the matching's survivors,
// 3: the pass-2 survivors of the stage pipeline).
int sf_launch_ba_pass(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass, int list_sel,
                      const uint8_t* mask, const uint8_t* run, bool fin, sf_result* d_out) {
  if (n <= 0) return SF_OK;
  if (sf_ba_lds_bytes(st.kcap) > 160 * 1024)
    return sf_fail(c, SF_ERANGE, "bundle adjustment needs %zu B of LDS (> 160 KiB)", sf_ba_lds_bytes(st.kcap));
  const bool pnp = c->dparams.estimation_type == 1;
  const int nw = c->ba_nw;
  int32_t* counters = (int32_t*)c->counters.p;
  const int32_t* list = (const int32_t*)(list_sel == 1 ? c->list1.p : c->list3.p);
  const int32_t* counter = counters + (list_sel == 1 ? 0 : 2);
  const uint32_t* corr = (const uint32_t*)(pass == 1 ? c->corr1.p : c->corr2.p);
  const CorrHeader* hdr = (const CorrHeader*)(pass == 1 ? c->hdr1.p : c->hdr2.p);
  PassState* ps = (PassState*)(pass == 1 ? c->pass1.p : c->pass2.p);
  const int end_3dof = c->dparams.force_3dof ? (pass == 1 ? 2 : 1) : 0;
  const bool both = st.kcap > SF_BA_SMALL_CAP;       // (a keyframe cannot give more words than it has features)
  sf_prof_begin(c, SF_K_BA);
  auto launch = [&](auto kern, int cap, bool& attr) -> int {
    const size_t lds = sf_ba_lds_bytes(cap);
    if (lds > 64 * 1024 && !attr) {
      SF_HIP(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(n), dim3(64 * nw), lds, c->stream, st, d_from, d_to, list, counter, corr, hdr, mask, run,
                       ps, end_3dof, fin ? 1 : 0, (const PassState*)c->pass1.p, (const uint8_t*)c->flags.p, d_out,
                       c->dparams, cap, both ? 1 : 0);
    return SF_OK;
  };
  int rc = SF_OK;
  bool dummy = true;      // (the SMALL launch's LDS is under 64 KB: no attribute to set)
  const int cap_small = std::min(st.kcap, SF_BA_SMALL_CAP);
#define SF_BA_CASE(NW_, PNP_)                                                                              \
  do {                                                                                                     \
    rc = (c->ba_occ ? c->ba_occ : (PNP_ ? 1 : 2)) == 2 ? launch(k_ba_pass<NW_, PNP_, true, 2>, cap_small, dummy)                          \
                        : launch(k_ba_pass<NW_, PNP_, true, 1>, cap_small, dummy);                         \
    if (rc == SF_OK && both) rc = launch(k_ba_pass<NW_, PNP_, false>, st.kcap, c->ba_pass_attr[PNP_][NW_ == 4 ? 2 : NW_ - 1]); \
  } while (0)
  if (pnp) { if (nw == 1) SF_BA_CASE(1, true); else if (nw == 2) SF_BA_CASE(2, true); else SF_BA_CASE(4, true); }
  else { if (nw == 1) SF_BA_CASE(1, false); else if (nw == 2) SF_BA_CASE(2, false); else SF_BA_CASE(4, false); }
#undef SF_BA_CASE
  sf_prof_end(c, SF_K_BA);
  if (rc != SF_OK) return rc;
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

// The split pipeline (k_match_split + k_chain); applies where the fused kernel does.
bool sf_split_applicable(const sf_context* c, const StoreView& st) {
  return sf_fused_lds_bytes(c, st) != 0 && c->match_mfma && st.kcap <= MF_MAX_ROWS &&
         (!c->dparams.bundle_adjustment || sf_ba_lds_bytes(st.kcap) <= 160 * 1024);
}

// the PnP form: k_match_split + k_chain_pnp
bool sf_split_pnp_applicable(const sf_context* c, const StoreView& st) {
  if (c->dparams.estimation_type != 1 || !c->fused || c->match_variant != 0 || c->params.desc_type != 0) return false;
  if (c->dparams.bidirectional) return false;      // (both directions: stage kernels)
  if (!(c->match_mfma && st.kcap <= MF_MAX_ROWS)) return false;
  const int nc = c->dparams.grid_gx * c->dparams.grid_gy;
  const size_t lds = std::max((sf_pnp_lds_bytes(st.kcap, c->dparams.iterations) + 15) & ~(size_t)15,
                              sf_guided_lds_bytes(st.kcap, nc));
  return lds <= 160 * 1024 && sf_match_lds_bytes(st.kcap, st.w) <= 160 * 1024 &&
         (!c->dparams.bundle_adjustment || sf_ba_lds_bytes(st.kcap) <= 160 * 1024);
}

namespace {
template <int W, int NW, int PART>
int launch_chain(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                 size_t lds, int tail_off, const BaHandover& H) {
  bool& attr_set = c->chain_attr[W == 16][PART][NW == 4 ? 2 : NW - 1];
  if (lds > 64 * 1024 && !attr_set) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_chain<W, NW, PART>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_chain<W, NW, PART>), dim3(n), dim3(64 * NW), lds, c->stream, st, d_from, d_to,
                     (const uint32_t*)c->corr1.p, (const CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p,
                     (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                     (const int32_t*)c->list1.p, (const int32_t*)(counters + 0), d_out, c->dparams, tail_off, H);
  return SF_OK;
}

template <int W, int PART, int NW>
int launch_chain_pnp(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                     const BaHandover& H) {
  const int nc = c->dparams.grid_gx * c->dparams.grid_gy;
  const size_t lds = std::max((sf_pnp_lds_bytes_dev(st.kcap, c->dparams.iterations, NW == 4, NW) + 15) & ~(size_t)15,
                              sf_guided_lds_bytes(st.kcap, nc, NW != 4));
  bool& attr = c->chain_pnp_attr[W == 16][PART][NW == 4 ? 2 : NW - 1];
  if (lds > 64 * 1024 && !attr) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_chain_pnp<W, PART, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int32_t* counters = (int32_t*)c->counters.p;
  hipLaunchKernelGGL((k_chain_pnp<W, PART, NW>), dim3(n), dim3(64 * NW), lds, c->stream, st, d_from, d_to,
                     (const uint32_t*)c->corr1.p, (const CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p,
                     (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                     (const int32_t*)c->list1.p, (const int32_t*)(counters + 0), d_out, c->dparams, H);
  return SF_OK;
}

template <int PART>
int launch_chain_part(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                      bool pnp, const BaHandover& H) {
  int rc;
  sf_prof_begin(c, SF_K_FUSED);
  if (pnp) {
#define SF_CHAIN_PNP_CASE(W_)                                                                            \
    rc = c->chain_pnp_nw == 1   ? launch_chain_pnp<W_, PART, 1>(c, st, d_from, d_to, n, d_out, H)          \
         : c->chain_pnp_nw == 2 ? launch_chain_pnp<W_, PART, 2>(c, st, d_from, d_to, n, d_out, H)          \
                                : launch_chain_pnp<W_, PART, 4>(c, st, d_from, d_to, n, d_out, H)
    if (st.w == 8) SF_CHAIN_PNP_CASE(8); else SF_CHAIN_PNP_CASE(16);
#undef SF_CHAIN_PNP_CASE
  } else {
    const int tail_off = (int)fused_tail_offset(c, st, false);
    const size_t lds_c = (size_t)tail_off + ((sizeof(FusedTail) + 15) & ~(size_t)15);
#define SF_CHAIN_CASE(W_)                                                                                             \
    rc = c->chain_nw == 1   ? launch_chain<W_, 1, PART>(c, st, d_from, d_to, n, d_out, lds_c, tail_off, H)            \
         : c->chain_nw == 2 ? launch_chain<W_, 2, PART>(c, st, d_from, d_to, n, d_out, lds_c, tail_off, H)            \
                            : launch_chain<W_, 4, PART>(c, st, d_from, d_to, n, d_out, lds_c, tail_off, H)
    if (st.w == 8) SF_CHAIN_CASE(8); else SF_CHAIN_CASE(16);
#undef SF_CHAIN_CASE
  }
  sf_prof_end(c, SF_K_FUSED);
  if (rc != SF_OK) return rc;
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
}  // namespace

int sf_launch_verify_split(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n,
                           sf_result* d_out) {
  if (n <= 0) return SF_OK;
  const bool pnp = c->dparams.estimation_type == 1;
  if (!(pnp ? sf_split_pnp_applicable(c, st) : sf_split_applicable(c, st)))
    return sf_fail(c, SF_EINVAL, "split verification pipeline not applicable");
  SF_HIP(c, hipMemsetAsync(c->counters.p, 0, 64, c->stream));
  // (Round 5, profiles/r05zc_matcher_workgroups_per_cu.txt: with its workgroups per CU capped by an inflated LDS request the
  //  launch alone takes 284 us at three per CU, 296 at two and 295 at ONE -- the pipelined scan keeps the matrix pipe busy
  //  from one wavefront per SIMD.  Capping it inside the overlapped step changes nothing (23.2-23.3 M pairs/s either way):
  //  the step as a whole is bound by vector instruction issue -- 425 M busy cycles of this kernel + 196 M of the chains +
  //  35 M of the NN kernels per step over 1 024 SIMDs = 0.32-0.36 ms of the 0.43 ms step, profiles/r05w_sq_*.)
  const size_t lds_m = sf_match_lds_bytes(st.kcap, st.w);
  int32_t* counters = (int32_t*)c->counters.p;
  sf_prof_begin(c, SF_K_MATCH);
  if (lds_m > 64 * 1024 && !c->split_match_attr[st.w == 16]) {
    if (st.w == 8)
      SF_HIP(c, hipFuncSetAttribute((const void*)k_match_split<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else
      SF_HIP(c, hipFuncSetAttribute((const void*)k_match_split<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    c->split_match_attr[st.w == 16] = true;
  }
#define SF_SPLIT_MATCH(...)                                                                                        \
  hipLaunchKernelGGL((k_match_split<__VA_ARGS__>), dim3(n), dim3(SF_BLOCK), lds_m, c->stream, st, d_from, d_to,    \
                     (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (CorrHeader*)c->hdr2.p, \
                     (PassState*)c->pass2.p, (uint8_t*)c->flags.p, (int32_t*)c->list1.p, counters + 0, d_out, c->dparams, \
                     sf_est_mode(c))
  // (shapes measured on the pipelined scan, profiles/r05t_match_split_shapes.txt: 4 tiles / 3 workgroups per CU 22.5 M
  //  pairs/s on the 3D-3D split step, 2 tiles / 4: 22.2, 2 tiles / 3: 21.3, 4 tiles / 2: 21.2; PnP 15.0 / 15.0 / 14.8 / 13.7)
  if (st.w == 8) SF_SPLIT_MATCH(8); else SF_SPLIT_MATCH(16);
#undef SF_SPLIT_MATCH
  sf_prof_end(c, SF_K_MATCH);
  SF_HIP(c, hipGetLastError());
  if (!c->dparams.bundle_adjustment)
    return launch_chain_part<0>(c, st, d_from, d_to, n, d_out, pnp, BaHandover{nullptr, nullptr, nullptr});
  // bundle adjustment on: [estimate 1] -> adjustment -> [guided matching + estimate 2] -> adjustment + result
  int rc;
  const size_t mb = (size_t)n * st.kcap;
  if ((rc = sf_buf_reserve(c, c->dir_mask, 2 * mb + (size_t)n)) != SF_OK) return rc;
  uint8_t* m1 = (uint8_t*)c->dir_mask.p;
  const BaHandover H = {m1, m1 + mb, m1 + 2 * mb};
  SF_HIP(c, hipMemsetAsync(m1, 0, 2 * mb + (size_t)n, c->stream));
  if ((rc = launch_chain_part<1>(c, st, d_from, d_to, n, d_out, pnp, H)) != SF_OK) return rc;
  if ((rc = sf_launch_ba_pass(c, st, d_from, d_to, n, 1, 1, H.mask1, nullptr, false, d_out)) != SF_OK) return rc;
  if ((rc = launch_chain_part<2>(c, st, d_from, d_to, n, d_out, pnp, H)) != SF_OK) return rc;
  return sf_launch_ba_pass(c, st, d_from, d_to, n, 2, 1, H.mask2, H.est2, true, d_out);
}
