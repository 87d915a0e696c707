// k_verify.hip -- the verification kernels as ONE translation unit, plus the fused per-pair pipeline.
//
// The three stage files are included (not linked) so that their per-pair bodies can be inlined into
// k_verify_fused: one 256-thread workgroup takes a candidate pair through the whole of
// StereoCamGeometricTools::estimateTransformation (stereoCamGeometricTools.cpp:122-178) --
// global matching, RANSAC, guess-guided matching, RANSAC again, result assembly -- without leaving the
// CU.  Why fuse: the motion-estimation stages are short dependent fp64 chains that occupy a few
// hundred workgroups for ~0.1 ms each and leave the issue ports idle, while matching is bound by
// instruction issue (matrix pipe + VALU top-2 scan; measured insensitive to 2 / 3 / 4 resident workgroups
// per CU).  In the fused kernel the 20 % of pairs that survive matching run their RANSAC chain while the
// other workgroups of the same CU are still matching.  Since matching moved to the matrix cores the two
// halves weigh about the same and the fused launch takes as long as the five stage launches (DESIGN.md
// section 5); it still is one launch instead of five for small batches.  The stage kernels remain for the
// PnP estimator (k_pnp: 177 VGPRs, 3 workgroups per CU) and as the A/B reference (SF_FUSED=0): both paths
// run the same bodies and produce identical bytes.
// Compiled with -ffp-contract=off (canonical arithmetic of the RANSAC / guided bodies).
#include "k_match.hip"
#include "k_ransac.hip"
#include "k_guided.hip"
#include "k_pnp.hip"

namespace {

// CW = wavefronts that run the motion-estimation chain of a surviving pair.  4 (default): the whole
// workgroup, as the stage kernels do.  1 or 2 (SF_CHAIN_WAVES / SF_OPT_CHAIN_WAVES): after matching the other
// wavefronts END and one or two carry the pair through RANSAC / guided matching / RANSAC
// (ransac_body<CW>, guided_body<W, CW>: same canonical sums, same integers, byte-identical results; the
// barriers inside the chain only see the live wavefronts).  The idea: a chain holds four wavefront slots and
// 4 x 128 VGPRs for ~100 us while using a fraction of one SIMD, and a CU whose four workgroup slots fill up
// with chains stops matching; ended wavefronts give slots and registers back.  Measured per 10 000 pairs:
// CW = 1 0.75 ms, CW = 2 0.68, CW = 4 0.59-0.61 -- a chain is more arithmetic than it looks (inlier counts
// of 64 hypotheses, rank counting for the median, the replayed 256-lane sums: ~200 us on one wavefront
// against ~106 on four), and the workgroup's LDS (26-30 KB, held until its last wavefront ends) caps a CU at
// five chains whatever their width.  Kept as options and as second implementations the tests compare against.
template <int W, int NQ, int CW>
__global__ void __launch_bounds__(SF_BLOCK, 4)
k_verify_fused(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
               uint32_t* __restrict__ corr1, CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
               uint32_t* __restrict__ corr2, CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2,
               uint8_t* __restrict__ guided_flag, sf_result* __restrict__ out, DeviceParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = blockIdx.x;
  SF_TRACE_MARK(P, pair, 0);
  // pass 1: global matching (myRegistrationVis.cpp:826-895) and, for survivors, RANSAC (:1113-1152)
  const bool est1 = match_v2_body<W, NQ, SF_BLOCK>(st, pair, pair_from, pair_to, P.nndr, P.min_inliers, 0, corr1, hdr1,
                                                   pass1, nullptr, nullptr, reinterpret_cast<int*>(smem_raw));
  __syncthreads();   // hdr1 / pass1 / corr1 of this pair are visible to the whole workgroup
  SF_TRACE_MARK(P, pair, 1);
  if constexpr (CW < 4) {
    // the wavefronts that stay rotate with the pair, so that the chains of a CU spread over its SIMDs
    if ((((threadIdx.x >> 6) - (unsigned)(pair & 3)) & 3u) >= (unsigned)CW) return;
  }
  // from here on this workgroup is a short chain of dependent fp64 steps: let its wavefronts win the
  // issue arbitration against the matching wavefronts it shares SIMDs with (they are throughput-bound
  // and lose nothing measurable), so the chain -- the tail of the launch -- finishes sooner
  __builtin_amdgcn_s_setprio(3);
  if (est1) {
    ransac_body<CW>(st, pair, pair_from, pair_to, corr1, hdr1, pass1, P, smem_raw);
    __syncthreads();
  }
  // pass 2: guess-guided matching (:476-825) seeded with the pass-1 pose, RANSAC again
  const bool est2 = guided_body<W, CW>(st, pair, pair_from, pair_to, pass1, pass2, guided_flag, corr2, hdr2, nullptr,
                                       nullptr, P, reinterpret_cast<int*>(smem_raw));
  __syncthreads();
  if (est2) {
    ransac_body<CW>(st, pair, pair_from, pair_to, corr2, hdr2, pass2, P, smem_raw, 11);
    __syncthreads();
  }
  SF_TRACE_MARK(P, pair, 17);
  if ((CW == 4 ? threadIdx.x : ((((threadIdx.x >> 6) - (unsigned)(pair & 3)) & 3u) * 64u + (threadIdx.x & 63u))) == 0)
    finalize_one(pair, pass1, pass2, guided_flag, out);
}

// The same pipeline with the PnP estimator (estimation_type = 1, myRegistrationVis.cpp:1055-1112), opt-in
// (SF_FUSED_PNP=1): k_pnp needs ~170 VGPRs, so three workgroups per CU, and a surviving pair's PnP chain is
// ~200 us (66 + 30 + 107) -- twice the 3D-3D one on fewer slots.  Measured three times, last with the matrix-core
// matcher: 0.95 ms per 10 000 pairs fused against 0.81 for the four stage launches.  Byte-identical either way
// (test_fused_pipeline_equals_stage_kernels[1]).
template <int W, int NQ>
__global__ void __launch_bounds__(SF_BLOCK, 3)
k_verify_fused_pnp(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
                   uint32_t* __restrict__ corr1, CorrHeader* __restrict__ hdr1, PassState* __restrict__ pass1,
                   uint32_t* __restrict__ corr2, CorrHeader* __restrict__ hdr2, PassState* __restrict__ pass2,
                   uint8_t* __restrict__ guided_flag, sf_result* __restrict__ out, DeviceParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pair = blockIdx.x;
  const int est = P.calibrated ? 1 : 2;     // gate selector of the matching body (sf_est_mode)
  const bool est1 = match_v2_body<W, NQ, SF_BLOCK>(st, pair, pair_from, pair_to, P.nndr, P.min_inliers, est, corr1, hdr1,
                                                   pass1, nullptr, nullptr, reinterpret_cast<int*>(smem_raw));
  __syncthreads();
  __builtin_amdgcn_s_setprio(3);
  if (est1) {
    pnp_body(st, pair, pair_from, pair_to, corr1, hdr1, pass1, P, smem_raw);
    __syncthreads();
  }
  const bool est2 = guided_body<W, 4>(st, pair, pair_from, pair_to, pass1, pass2, guided_flag, corr2, hdr2, nullptr,
                                      nullptr, P, reinterpret_cast<int*>(smem_raw));
  __syncthreads();
  if (est2) {
    pnp_body(st, pair, pair_from, pair_to, corr2, hdr2, pass2, P, smem_raw);
    __syncthreads();
  }
  if (threadIdx.x == 0) finalize_one(pair, pass1, pass2, guided_flag, out);
}

template <int W, int NQ>
int launch_fused_pnp(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                     size_t lds) {
  bool& attr_set = c->fused_pnp_attr[W == 16][NQ == 0];
  if (lds > 64 * 1024 && !attr_set) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_verify_fused_pnp<W, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_verify_fused_pnp<W, NQ>), dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                     (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (uint32_t*)c->corr2.p,
                     (CorrHeader*)c->hdr2.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p, d_out, c->dparams);
  return SF_OK;
}

template <int W, int NQ, int CW>
int launch_fused(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, sf_result* d_out,
                 size_t lds) {
  bool& attr_set = c->fused_attr[W == 16][NQ == 0][CW == 4 ? 0 : CW];   // one flag per instantiation
  if (lds > 64 * 1024 && !attr_set) {
    SF_HIP(c, hipFuncSetAttribute((const void*)k_verify_fused<W, NQ, CW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_verify_fused<W, NQ, CW>), dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                     (uint32_t*)c->corr1.p, (CorrHeader*)c->hdr1.p, (PassState*)c->pass1.p, (uint32_t*)c->corr2.p,
                     (CorrHeader*)c->hdr2.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p, d_out, c->dparams);
  return SF_OK;
}

}  // namespace

// Dynamic LDS of the fused kernel = the largest stage; 0 when the fused pipeline does not apply
// (PnP estimator, SF_FUSED=0, or a stage that needs more than the 160 KB of a CU).
size_t sf_fused_lds_bytes(const sf_context* c, const StoreView& st) {
  if (!c->fused || c->dparams.estimation_type > 1) return 0;
  if (c->dparams.estimation_type == 1 && !c->fused_pnp) return 0;
  const int nc = c->dparams.grid_gx * c->dparams.grid_gy;
  const size_t match = (size_t)(st.kcap * st.w + 2 * st.kcap + 16) * sizeof(int);
  const size_t guided = sf_guided_lds_bytes(st.kcap, nc);
  const size_t ransac = c->dparams.estimation_type == 1 ? sf_pnp_lds_bytes(st.kcap, c->dparams.iterations)
                                                        : sf_ransac_lds_bytes(st.kcap, c->dparams.iterations);
  // same rule as sf_launch_match_global: the LDS-staged matching body only while the staged "from"
  // block leaves room for >= 2 workgroups per CU; beyond that the stage kernels (scalar-load matcher)
  if (match > 64 * 1024 || c->match_variant != 0) return 0;
  const size_t lds = std::max(match, std::max(guided, ransac));
  return lds <= 160 * 1024 ? lds : 0;
}

int sf_launch_verify_fused(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n,
                           sf_result* d_out) {
  if (n <= 0) return SF_OK;
  const size_t lds = sf_fused_lds_bytes(c, st);
  if (lds == 0) return sf_fail(c, SF_EINVAL, "fused verification pipeline not applicable");
  int rc;
  sf_prof_begin(c, SF_K_FUSED);
  const bool mf = c->match_mfma && st.kcap <= MF_MAX_ROWS;
  if (c->dparams.estimation_type == 1) {
    if (mf) rc = st.w == 8 ? launch_fused_pnp<8, 0>(c, st, d_from, d_to, n, d_out, lds)
                           : launch_fused_pnp<16, 0>(c, st, d_from, d_to, n, d_out, lds);
    else rc = st.w == 8 ? launch_fused_pnp<8, 2>(c, st, d_from, d_to, n, d_out, lds)
                        : launch_fused_pnp<16, 2>(c, st, d_from, d_to, n, d_out, lds);
    sf_prof_end(c, SF_K_FUSED);
    if (rc != SF_OK) return rc;
    SF_HIP(c, hipGetLastError());
    return SF_OK;
  }
#define SF_FUSED_CASE(W_, NQ_)                                                                    \
  rc = c->chain_waves == 1   ? launch_fused<W_, NQ_, 1>(c, st, d_from, d_to, n, d_out, lds)        \
       : c->chain_waves == 2 ? launch_fused<W_, NQ_, 2>(c, st, d_from, d_to, n, d_out, lds)        \
                             : launch_fused<W_, NQ_, 4>(c, st, d_from, d_to, n, d_out, lds)
  if (mf) {
    if (st.w == 8) SF_FUSED_CASE(8, 0); else SF_FUSED_CASE(16, 0);
  } else {
    if (st.w == 8) SF_FUSED_CASE(8, 2); else SF_FUSED_CASE(16, 2);
  }
#undef SF_FUSED_CASE
  sf_prof_end(c, SF_K_FUSED);
  if (rc != SF_OK) return rc;
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
