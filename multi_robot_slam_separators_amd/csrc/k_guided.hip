// k_guided.hip -- pass-2 GUESS-GUIDED window matching, one candidate pair per 256-thread workgroup.
//
// Replaces myRegistrationVis.cpp:476-825 of the reference (default sub-branch :667-818,
// _guessMatchToProjection = false): project the "from" 3D points into the "to" image with the
// pass-1 pose, search the "to" keypoints inside a guess_win_size-pixel radius around every
// projection, keep candidates of the same octave, brute-force Hamming k=2 + NNDR among them (a
// single candidate is accepted without descriptor test), every "to" keypoint claimed once by the
// lowest "from" index.  FLANN's approximate kd-tree radius search [upstream] is replaced by an
// EXACT scan of the K_to keypoints.
//
// CDNA4 mapping: one lane per "from" point (projection in fp64 as cv::projectPoints does); the
// "to" keypoints {x, y, octave} are wave-uniform float4 records fetched with scalar loads; the
// Hamming distance is only evaluated for the few (from, to) combinations that pass the window
// and octave tests; claims use LDS atomicMin; the id-ordered compaction uses wavefront ballots.
// Also the stage that decides what pass 2 is for every pair (stereoCamGeometricTools.cpp:153-164):
// a failed pass 1 or an unusable guess re-runs the global branch, whose deterministic result is
// the pass-1 state, so those pairs are finished here without re-computation.
// Compiled with -ffp-contract=off (canonical arithmetic).
#include "sf_device_math.hpp"
#include "sf_internal.hpp"

namespace {

// recorded (from, to) combinations per thread of the candidate-parallel search (list = GUIDED_CPT * 256 entries); a
// frame with more takes the per-lane loop
#define GUIDED_CPT 8
// ... and the list's length: the same for every workgroup width (a narrower workgroup walks it in more trips)
#define GUIDED_CAND_CAP (GUIDED_CPT * 256)
// The narrow chains (one or two wavefronts per workgroup) hold a shorter list -- 1 216 combinations: the bench's frames
// have 760 on average, 1 080 at the 90th percentile -- so that the guided pass fits six PnP chains to a CU (26.4 KB at
// K = 500); a frame with more takes the per-lane loop (same integers).
#define GUIDED_CAND_CAP_NARROW 1216

// Body of the pass-2 matching stage for ONE pair (the calling workgroup); returns whether the pair
// needs the pass-2 motion estimation (block-uniform).  list == nullptr: no work-list append (fused).
// `out` / hdr_out / pass2_out / guided_flag_out: the pair's pass-2 correspondence list, header, state and flag
// (global arrays in the stage kernel, LDS in the fused kernel).
// L2: float32 descriptor rows of W dimensions (desc_type 1): the combinations that pass the window and octave tests are
// compared by their L2 distance -- sqrtf of the float32 sum of squared differences in dimension order, as
// cv::BFMatcher(NORM_L2) reports it (:739-749) -- in the per-lane search loop (same tests, same decisions; the
// candidate-parallel pass with its packed 32-bit keys is the binary descriptors').
template <int W, bool L2 = false, int NW = 4>
__device__ __forceinline__ bool guided_body(const StoreView& st, int pair, int sF, int sT, const PassState& pass1,
                                            PassState& pass2_out, uint8_t& guided_flag_out, uint32_t* out,
                                            CorrHeader& hdr_out, int32_t* __restrict__ list,
                                            int32_t* __restrict__ counter, const DeviceParams& P, int* smem) {
  constexpr int NT = 64 * NW;      // NW = 4: the 256-thread workgroups; 1, 2: the low-occupancy chains (k_verify.hip)
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int kcap = st.kcap;
  const PassState p1 = pass1;
  const bool bad_slot = (unsigned)sF >= (unsigned)st.n_slots || (unsigned)sT >= (unsigned)st.n_slots;
  const int4 mF = bad_slot ? make_int4(0, 0, 0, 0) : st.meta[sF];
  const int4 mT = bad_slot ? make_int4(0, 0, 0, 0) : st.meta[sT];
  const int Kf = mF.x, Kt = mT.x;

  bool ident = true;
#pragma unroll
  for (int i = 0; i < 12; ++i) ident = ident && (p1.T[i] == ((i == 0 || i == 5 || i == 10) ? 1.f : 0.f));
  // myRegistrationVis.cpp:477-479
  const bool eligible = !p1.is_null && !ident && P.guess_win > 0 && mF.y > 0 && P.calibrated && Kf > 0 && Kt > 0;
  if (!eligible) {
    if (tid == 0) {
      pass2_out = p1;
      guided_flag_out = 0;
      CorrHeader h = {0, 0, 0, 0};
      hdr_out = h;
    }
    return false;
  }

  int* claim = smem;               // [kcap] lowest "from" index that matched each "to" row
  int* matched = smem + kcap;      // [kcap] "to" row matched by each "from" point, or -1
  int* misc = smem + 2 * kcap;     // [16]
  const int NC = P.grid_gx * P.grid_gy;
  int* cell_start = misc + 16;     // [NC + 1] CSR of "to" keypoints bucketed on a uniform grid
  // [NC] fill counters of the bucketing: only alive until the keypoints are in their cells, so they live in `matched`
  // (written by the decisions at the very end) when that is large enough -- round 3: the guided pass sizes the fused
  // kernel's LDS, and K = 1000 frames were 10 KB over what lets three workgroups share a CU
  // Round 5: up to 2 kcap cells they live in `claim` + `matched` (contiguous; `claim` is initialised behind the bucketing
  // now): the 768 cells of a 640 x 480 image over K = 500 frames cost 3 KB of their own -- what kept the two-wavefront PnP
  // chain at five per CU instead of six.
  const bool fill_in_matched = NC <= 2 * kcap;
  int* cell_fill = fill_in_matched ? claim : cell_start + NC + 1;
  // [kcap] the "to" keypoints grouped by cell as {x, y, octave bits, index bits}: the window test never leaves
  // LDS and takes ONE 16-byte read per entry
  // (the 16-byte alignment is done on the int INDEX: rounding the pointer through uintptr_t, as round 1 did, hid the
  //  LDS address space from the compiler and turned every access behind it -- the window tests' entry reads, the
  //  key atomics -- into FLAT instructions)
  float4* item4 = reinterpret_cast<float4*>(smem + ((2 * kcap + 16 + NC + 1 + (fill_in_matched ? 0 : NC) + 3) & ~3));
  // (a point's projection is not kept: the per-lane search behind a candidate-list overflow recomputes it)
  uint32_t* oilast = reinterpret_cast<uint32_t*>(item4 + kcap);  // [kcap] candidates of the point << 16 | highest one
  uint32_t* key1 = oilast + kcap;                                // [kcap] best (Hamming << 16 | to) of the point
  uint32_t* key2 = key1 + kcap;                                  // [kcap] second best
  uint32_t* cand = key2 + kcap;                                  // [GUIDED_CPT * 256] recorded combinations (from << 16 | to)
  for (int i = tid; i <= NC; i += NT) cell_start[i] = 0;
  for (int i = tid; i < NC; i += NT) cell_fill[i] = 0;
  if (tid < 16) misc[tid] = 0;

  // :486-487 guessCameraRef = (guess * localTransform).inverse()
  float Rc[9], tc[3];
  {
    const float* g = p1.T;
    const float* Lt = P.L;
    float GR[9], Gt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        GR[3 * i + j] = (g[4 * i] * Lt[j] + g[4 * i + 1] * Lt[4 + j]) + g[4 * i + 2] * Lt[8 + j];
      Gt[i] = ((g[4 * i] * Lt[3] + g[4 * i + 1] * Lt[7]) + g[4 * i + 2] * Lt[11]) + g[4 * i + 3];
    }
    // Every coefficient goes through an (empty) register barrier: the twelve values must not be packed two to a
    // v_pk_*_f32 instruction.  Round 3 found about one survivor chain in a thousand of k_verify_fused projecting the
    // points of lanes 48-63 of one wavefront with different X coefficients (the LOW halves of the packed results; Y and
    // Z, the high halves and the scalar ones, were right) when this block was SLP-vectorised -- DESIGN.md section 3.
    // Kept scalar here AND the translation unit is built with -fno-slp-vectorize; either alone removed the symptom.
#ifndef SF_NO_PK_BARRIERS      // (tools/pk_isa_scan.py builds the block WITHOUT them, SLP on, to show the instructions of the symptom)
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(GR[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(Gt[i]));
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rc[3 * i + j] = GR[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tc[i] = -((Rc[3 * i] * Gt[0] + Rc[3 * i + 1] * Gt[1]) + Rc[3 * i + 2] * Gt[2]);
#ifndef SF_NO_PK_BARRIERS
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(tc[i]));
#endif
  }
  __syncthreads();

  const uint32_t* dF = st.desc + (size_t)sF * kcap * W;
  const uint32_t* dT = st.desc + (size_t)sT * kcap * W;

  const float* xF = st.xyz + (size_t)sF * kcap * 3;
  const float4* kF = st.kp + (size_t)sF * kcap;
  const float4* kT = st.kp + (size_t)sT * kcap;
  const float r2lim = (float)P.guess_win * (float)P.guess_win;

  // ---- bucket the "to" keypoints: the reference searches a kd-tree (myRegistrationVis.cpp:670-680);
  // a uniform grid with cell >= window radius makes the exact radius search a 3x3-cell visit.
  // Keypoints outside the image are clamped into border cells (any point within the radius of an
  // in-image projection still lands in that projection's 3x3 neighbourhood); non-finite ones can
  // never pass the window test and are left out.
  const float inv_cell = P.grid_inv_cell;
  const int gxm = P.grid_gx - 1, gym = P.grid_gy - 1;
  const float reach = (float)P.guess_win * 1.0001f + 1e-3f;   // window radius with a rounding margin
  // A keypoint farther than the window radius outside the image cannot be within the radius of any in-image
  // projection (only those are searched, :503-512): it is not bucketed at all.  Clamping such points into the
  // border cells instead made every search near a border walk through all of them (synthetic frames whose
  // features fill a wider field of view than the image put a third of their keypoints there).
  const float x_hi = P.wlim + reach, y_hi = P.hlim + reach;
  // All of this stage's streaming loads are issued up front, two per lane and array: the "to" keypoints (staged raw
  // in LDS for the second bucketing pass -- the region is the search's scratch, unused until then) and the first
  // 512 "from" points with their octaves (kept in registers until the search).  Round 1 re-read the keypoints from
  // HBM for the fill pass and loaded the "from" points inside the search loop.
  float4* raw = reinterpret_cast<float4*>(oilast);      // [kcap] (aliases oilast / key1 / key2 / cand: 3 kcap + 2048 words >= 4 kcap)
  float pre_x[2], pre_y[2], pre_z[2];
  int pre_o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + j * NT;
    pre_x[j] = pre_y[j] = pre_z[j] = 0.f;
    pre_o[j] = 0;
    if (i < Kf) {
      pre_x[j] = xF[3 * i]; pre_y[j] = xF[3 * i + 1]; pre_z[j] = xF[3 * i + 2];
      pre_o[j] = __float_as_int(kF[i].z);
    }
  }
  auto cell_of = [&](const float4& k) -> int {
    if (isfinite(k.x) && isfinite(k.y) && k.x >= -reach && k.x < x_hi && k.y >= -reach && k.y < y_hi) {
      const int cx = min(max((int)floorf(fminf(fmaxf(k.x * inv_cell, -1.f), 1e6f)), 0), gxm);
      const int cy = min(max((int)floorf(fminf(fmaxf(k.y * inv_cell, -1.f), 1e6f)), 0), gym);
      return cy * P.grid_gx + cx;
    }
    return -1;
  };
  for (int base = 0; base < Kt; base += 2 * NT) {
    float4 k2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = base + tid + j * NT;
      k2[j] = make_float4(__int_as_float(0x7FC00000), 0.f, 0.f, 0.f);
      if (t < Kt) k2[j] = kT[t];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = base + tid + j * NT;
      if (t < Kt) {
        raw[t] = k2[j];
        const int cidx = cell_of(k2[j]);
        if (cidx >= 0) atomicAdd(&cell_start[cidx + 1], 1);
      }
    }
  }
  __syncthreads();
  {
    // exclusive scan of the per-cell counts (each thread owns a run of consecutive cells)
    const int per = (NC + NT - 1) / NT;
    const int c0 = tid * per, c1 = min(c0 + per, NC);
    int local = 0;
    for (int cidx = c0; cidx < c1; ++cidx) local += cell_start[cidx + 1];
    int incl = local;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) misc[8 + wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += misc[8 + w];
    int run = woff + incl - local;
    __syncthreads();   // all counts read before they are overwritten with offsets
    for (int cidx = c0; cidx < c1; ++cidx) {
      const int cnt_c = cell_start[cidx + 1];
      cell_start[cidx + 1] = run + cnt_c;     // becomes the END offset of cell cidx = start of cidx + 1
      run += cnt_c;
    }
  }
  __syncthreads();
  for (int t = tid; t < Kt; t += NT) {
    const float4 k = raw[t];
    const int cidx = cell_of(k);
    if (cidx >= 0) {
      const int pos = cell_start[cidx] + atomicAdd(&cell_fill[cidx], 1);
      item4[pos] = make_float4(k.x, k.y, k.z, __int_as_float(t));
    }
  }
  __syncthreads();
  SF_TRACE_MARK(P, pair, 8);

  // ---- search.  Two formulations of the same integers:
  //  * candidate-parallel (default): every lane projects its "from" point and walks the 3x3 cells around the
  //    projection, but only RECORDS the (from, to) combinations that pass the window and octave tests; the
  //    Hamming distances of all recorded combinations are then evaluated one combination per lane -- the two
  //    descriptor loads of every combination are in flight together, instead of one dependent HBM round trip per
  //    combination inside a divergent per-lane loop -- and folded into per-point best / second-best keys with LDS
  //    atomics;
  //  * sequential per lane (the round-1 loop): when a frame has more combinations than the list holds
  //    (GUIDED_CPT per thread), e.g. all keypoints piled on one spot.
#ifndef SF_CHAIN_TRACE
  if (P.dbg_trace) {
    // experiment: is the bucketing complete when this wavefront starts to search?  (a wavefront that is one barrier
    // ahead of the others would see cells whose fill count is short of their size)
    int short_cells = 0;
    for (int c0 = tid; c0 < NC; c0 += NT) short_cells += (cell_fill[c0] != cell_start[c0 + 1] - cell_start[c0]) ? 1 : 0;
    if (short_cells) atomicAdd(&P.dbg_trace[0], (unsigned long long)short_cells);
    if (tid == 0) atomicAdd(&P.dbg_trace[1], 1ull);
    __syncthreads();                                       // (the counters are about to be overwritten)
  }
#endif
  for (int i = tid; i < Kf; i += NT) matched[i] = -1;      // (the fill counters are dead; ordered before the decisions by
  for (int i = tid; i < Kt; i += NT) claim[i] = 0x7FFFFFFF;  //  the barrier behind the search's first pass)
  int n_finite = 0, n_proj = 0;
  const int cand_cap = NW == 4 ? GUIDED_CAND_CAP : min(GUIDED_CAND_CAP_NARROW, 4 * kcap);   // (NW < 4: the keys are parked in the grid's items)
  // projection of a finite "from" point with the guess (:503-512): pixel position and "inside the image, in front"
  auto project = [&](float x, float y, float z, float& u, float& v) -> bool {
    const float zf = ((Rc[6] * x + Rc[7] * y) + Rc[8] * z) + tc[2];
    const double X = (((double)Rc[0] * (double)x + (double)Rc[1] * (double)y) + (double)Rc[2] * (double)z) + (double)tc[0];
    const double Y = (((double)Rc[3] * (double)x + (double)Rc[4] * (double)y) + (double)Rc[5] * (double)z) + (double)tc[1];
    const double Z = (((double)Rc[6] * (double)x + (double)Rc[7] * (double)y) + (double)Rc[8] * (double)z) + (double)tc[2];
    const double iz = (Z != 0.0) ? 1.0 / Z : 1.0;
    u = (float)((X * iz) * P.fx + P.cx);
    v = (float)((Y * iz) * P.fy + P.cy);
    return isfinite(u) && isfinite(v) && !(u < 0.f) && !(u >= P.wlim) && !(v < 0.f) && !(v >= P.hlim) && (zf > 0.f);
  };
  // pass A: projections, per-point candidate count / highest candidate (:751-764 needs it), candidate list
  for (int base = 0; base < Kf; base += NT) {
    const int i = base + tid;
    bool inimg = false;
    float u = 0.f, v = 0.f;
    const int rnd = base / NT;
    float x = 0.f, y = 0.f, z = 0.f;
    int octf = 0;
    if (rnd < 2) {               // (the first 512 points were loaded up front; larger frames load here)
      x = rnd == 0 ? pre_x[0] : pre_x[1]; y = rnd == 0 ? pre_y[0] : pre_y[1]; z = rnd == 0 ? pre_z[0] : pre_z[1];
      octf = rnd == 0 ? pre_o[0] : pre_o[1];
    } else if (i < Kf) {
      x = xF[3 * i]; y = xF[3 * i + 1]; z = xF[3 * i + 2];
      octf = __float_as_int(kF[i].z);
    }
    if (i < Kf) {
      if (sfd::finite3(x, y, z)) {
        ++n_finite;
        inimg = project(x, y, z, u, v);
      }
    }
    // Window / octave tests of this lane's point against the 3x3 cells around its projection.  Combinations that
    // pass are parked in four registers and handed to the workgroup's list in bulk: one slot allocation per
    // WAVEFRONT and round (prefix sum of the per-lane counts + one LDS atomic by a single lane), plus a per-lane
    // flush for the few points with more than four.
    int oi = 0, last = -1, nb = 0;
    uint32_t hb0 = 0, hb1 = 0, hb2 = 0, hb3 = 0;
#ifdef SF_CHAIN_TRACE
    int dbg_trips = 0, dbg_entries = 0;
#endif
    if (base == 0) SF_TRACE_MARK(P, pair, 28);   // (thread 0's own timeline: first round projected)
    if (i < Kf && inimg) {
      ++n_proj;
      const int cx0 = min(max((int)floorf((u - reach) * inv_cell), 0), gxm);
      const int cx1 = min(max((int)floorf((u + reach) * inv_cell), 0), gxm);
      const int cy0 = min(max((int)floorf((v - reach) * inv_cell), 0), gym);
      const int cy1 = min(max((int)floorf((v + reach) * inv_cell), 0), gym);
      for (int cy = cy0; cy <= cy1; ++cy) {
        // cells cx0..cx1 of one grid row are contiguous in the CSR; entries are read FOUR at a time (the loop is
        // one LDS round trip per trip, and a wavefront runs as many trips as its busiest lane)
        const int e0 = cell_start[cy * P.grid_gx + cx0], e1 = cell_start[cy * P.grid_gx + cx1 + 1];
        for (int e = e0; e < e1; e += 4) {
#ifdef SF_CHAIN_TRACE
          ++dbg_trips; dbg_entries += min(4, e1 - e);
#endif
          float4 it4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) it4[q] = item4[min(e + q, e1 - 1)];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 it = it4[q];
            const float dx = u - it.x, dy = v - it.y;
            const float d2 = dx * dx + dy * dy;
            if (e + q < e1 && d2 < r2lim && __float_as_int(it.z) == octf) {
              const int t = __float_as_int(it.w);
              if (nb == 4) {                     // (rare) more than four for one point: flush this lane's four
                const int slot = atomicAdd(&misc[3], 4);
                if (slot + 4 <= cand_cap) {
                  cand[slot] = hb0; cand[slot + 1] = hb1; cand[slot + 2] = hb2; cand[slot + 3] = hb3;
                }
                nb = 0;
              }
              const uint32_t rec = ((uint32_t)i << 16) | (uint32_t)t;
              hb0 = nb == 0 ? rec : hb0; hb1 = nb == 1 ? rec : hb1; hb2 = nb == 2 ? rec : hb2; hb3 = nb == 3 ? rec : hb3;
              ++nb;
              ++oi;
              if (last < 0 || t > last) last = t;
            }
          }
        }
      }
    }
#ifdef SF_CHAIN_TRACE
    if (base == 0 && wave == 0 && P.dbg_trace) {
      int mx = dbg_trips, sm = dbg_entries;
      for (int off = 32; off >= 1; off >>= 1) { mx = max(mx, __shfl_xor(mx, off)); sm += __shfl_xor(sm, off); }
      if (lane == 0) P.dbg_trace[(size_t)pair * SF_TRACE_SLOTS + 31] = ((unsigned long long)mx << 32) | (unsigned)sm;
    }
#endif
    if (base == 0) SF_TRACE_MARK(P, pair, 29);   // ... its cells scanned
    {
      // bulk hand-over of the parked combinations of this wavefront
      int incl = nb;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const int wtot = __shfl(incl, 63);
      int wbase = 0;
      if (wtot > 0) {
        if (lane == 0) wbase = atomicAdd(&misc[3], wtot);
        wbase = __shfl(wbase, 0);
        const int s0 = wbase + incl - nb;
        if (s0 + nb <= cand_cap) {
          if (nb > 0) cand[s0] = hb0;
          if (nb > 1) cand[s0 + 1] = hb1;
          if (nb > 2) cand[s0 + 2] = hb2;
          if (nb > 3) cand[s0 + 3] = hb3;
        }
      }
    }
    if (base == 0) SF_TRACE_MARK(P, pair, 30);   // ... combinations handed over
#ifndef SF_CHAIN_TRACE
    if (P.dbg_trace && i < Kf) {      // experiment: what this lane projected and found (sf_debug_guided_points)
      unsigned long long* pl0 = P.dbg_trace + 512 + (size_t)pair * kcap;
      unsigned long long* pl1 = pl0 + (size_t)gridDim.x * kcap;
      pl0[i] = ((unsigned long long)__float_as_uint(u) << 32) | (unsigned long long)__float_as_uint(v);
      pl1[i] = ((unsigned long long)(unsigned)oi << 48) | ((unsigned long long)(unsigned)(last & 0xFFFF) << 32) |
               ((unsigned long long)(inimg ? 1u : 0u) << 31) | (unsigned long long)((unsigned)octf & 0xFFFFu);
    }
#endif
    if (i < Kf) {
      oilast[i] = ((uint32_t)min(oi, 0xFFFF) << 16) | (uint32_t)(last & 0xFFFF);
      key1[i] = 0xFFFFFFFFu;
      key2[i] = 0xFFFFFFFFu;
    }
  }
  __syncthreads();
  SF_TRACE_MARK(P, pair, 24);     // projections + candidate recording done
  const int n_cand = misc[3];
#ifndef SF_CHAIN_TRACE
  if (P.dbg_trace) {
    // every point's candidate count (oilast >> 16, written by its lane) must add up to the list's length
    int mine = 0;
    for (int i = tid; i < Kf; i += NT) mine += (int)(oilast[i] >> 16);
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) atomicAdd(&misc[12], mine);        // misc[12..]: free words
    __syncthreads();
    if (tid == 0 && misc[12] != n_cand) atomicAdd(&P.dbg_trace[2], 1ull);
    __syncthreads();
  }
#endif
#ifdef SF_CHAIN_TRACE
  if (tid == 0 && P.dbg_trace) P.dbg_trace[(size_t)pair * SF_TRACE_SLOTS + 27] = (unsigned long long)n_cand;
#endif
  if (!L2 && n_cand <= cand_cap) {
    if constexpr (NW == 4) {
    // pass B: one combination per lane (<= GUIDED_CPT per thread, kept in registers between the two atomic passes)
    uint32_t ck[GUIDED_CPT], ci[GUIDED_CPT];
#pragma unroll
    for (int j = 0; j < GUIDED_CPT; ++j) {
      const int c = tid + j * NT;
      ck[j] = 0xFFFFFFFFu;
      ci[j] = 0;
      if (c < n_cand) {
        const uint32_t it = cand[c];
        const uint32_t i = it >> 16, t = it & 0xFFFFu;
        const uint4* pf = reinterpret_cast<const uint4*>(dF + (size_t)i * W);
        const uint4* pt = reinterpret_cast<const uint4*>(dT + (size_t)t * W);
        uint32_t d = 0;
#pragma unroll
        for (int q4 = 0; q4 < W / 4; ++q4) {
          const uint4 a = pf[q4], b = pt[q4];
          d += __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
        }
        ck[j] = (d << 16) | t;
        ci[j] = i;
        atomicMin(&key1[i], ck[j]);
      }
    }
    __syncthreads();
    SF_TRACE_MARK(P, pair, 25);   // Hamming distances + best keys
#pragma unroll
    for (int j = 0; j < GUIDED_CPT; ++j)
      if (ck[j] != 0xFFFFFFFFu && ck[j] != key1[ci[j]]) atomicMin(&key2[ci[j]], ck[j]);   // keys of a point are distinct
    __syncthreads();
    } else {
    // pass B on one or two wavefronts: the same keys, but a lane takes GUIDED_CAND_CAP / NT combinations and parks their
    // keys in LDS -- the grid's items (4 kcap words) are dead once the candidates are recorded -- instead of in registers;
    // U combinations per trip, so that their descriptor loads are in flight together
    constexpr int U = 2;
    uint32_t* ckey = reinterpret_cast<uint32_t*>(item4);
    for (int c0 = U * tid; c0 < n_cand; c0 += U * NT) {
      uint32_t ck[U], ci[U];
      uint4 fa[U][W / 4], fb[U][W / 4];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t it = cand[min(c0 + j, n_cand - 1)];
        ci[j] = it >> 16;
        ck[j] = it & 0xFFFFu;
        const uint4* pf = reinterpret_cast<const uint4*>(dF + (size_t)ci[j] * W);
        const uint4* pt = reinterpret_cast<const uint4*>(dT + (size_t)ck[j] * W);
#pragma unroll
        for (int q4 = 0; q4 < W / 4; ++q4) { fa[j][q4] = pf[q4]; fb[j][q4] = pt[q4]; }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        uint32_t d = 0;
#pragma unroll
        for (int q4 = 0; q4 < W / 4; ++q4) {
          const uint4 a = fa[j][q4], b = fb[j][q4];
          d += __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
        }
        if (c0 + j < n_cand) {
          const uint32_t key = (d << 16) | ck[j];
          ckey[c0 + j] = key;
          atomicMin(&key1[ci[j]], key);
        }
      }
    }
    __syncthreads();
    SF_TRACE_MARK(P, pair, 25);   // Hamming distances + best keys
    for (int c = tid; c < n_cand; c += NT) {
      const uint32_t key = ckey[c], i = cand[c] >> 16;
      if (key != key1[i]) atomicMin(&key2[i], key);   // keys of a point are distinct
    }
    __syncthreads();
    }
    SF_TRACE_MARK(P, pair, 26);   // second-best keys
    for (int i = tid; i < Kf; i += NT) {
      const uint32_t ol = oilast[i];
      const int oi = (int)(ol >> 16);
      int m = -1;
      if (oi >= 2) {
        const uint32_t b0 = key1[i], b1 = key2[i];
        if ((float)(b0 >> 16) < P.nndr * (float)(b1 >> 16)) m = (int)(b0 & 0xFFFFu);  // :744
      } else if (oi == 1) {
        m = (int)(ol & 0xFFFFu);                                                     // :751-764
      }
      if (m >= 0) {
        matched[i] = m;
        atomicMin(&claim[m], i);                                                      // :776-787
      }
    }
  } else {
    // more combinations than the list holds: the per-lane loop (same tests, same keys)
    for (int i = tid; i < Kf; i += NT) {
      const float px = xF[3 * i], py = xF[3 * i + 1], pz = xF[3 * i + 2];
      float u = 0.f, v = 0.f;
      if (!sfd::finite3(px, py, pz) || !project(px, py, pz, u, v)) continue;     // not searched
      const int octf = __float_as_int(kF[i].z);
      uint32_t q[W];
      {
        const uint4* p = reinterpret_cast<const uint4*>(dF + (size_t)i * W);
#pragma unroll
        for (int c = 0; c < W / 4; ++c) {
          uint4 t4 = p[c];
          q[4 * c] = t4.x; q[4 * c + 1] = t4.y; q[4 * c + 2] = t4.z; q[4 * c + 3] = t4.w;
        }
      }
      int oi = 0, last = -1;
      uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;  // (dist << 16 | to_idx) keys
      float f0 = __int_as_float(0x7F800000), f1 = __int_as_float(0x7F800000);   // L2: best / second-best distance ...
      int t0 = -1;                                                               // ... and the best's "to" row
      const int cx0 = min(max((int)floorf((u - reach) * inv_cell), 0), gxm);
      const int cx1 = min(max((int)floorf((u + reach) * inv_cell), 0), gxm);
      const int cy0 = min(max((int)floorf((v - reach) * inv_cell), 0), gym);
      const int cy1 = min(max((int)floorf((v + reach) * inv_cell), 0), gym);
      for (int cy = cy0; cy <= cy1; ++cy) {
        const int e0 = cell_start[cy * P.grid_gx + cx0], e1 = cell_start[cy * P.grid_gx + cx1 + 1];
        for (int e = e0; e < e1; ++e) {
          const float4 it = item4[e];
          const float dx = u - it.x, dy = v - it.y;
          const float d2 = dx * dx + dy * dy;
          if (d2 < r2lim && __float_as_int(it.z) == octf) {
            const int t = __float_as_int(it.w);
            const uint32_t* r = dT + (size_t)t * W;
            if constexpr (L2) {
              float sq = 0.f;
#pragma unroll
              for (int c = 0; c < W; ++c) {
                const float dd = __uint_as_float(q[c]) - __uint_as_float(r[c]);
                sq = sq + dd * dd;
              }
              const float d = sqrtf(sq);
              // (candidates arrive in grid order, not in "to" order: the better of two equal distances is the lower row,
              //  as a scan in "to" order with strict comparisons leaves it)
              if (d < f0 || (d == f0 && t < t0)) { f1 = f0; f0 = d; t0 = t; }
              else if (d < f1) { f1 = d; }
            } else {
              uint32_t d = 0;
#pragma unroll
              for (int c = 0; c < W; ++c) d += __popc(r[c] ^ q[c]);
              const uint32_t key = (d << 16) | (uint32_t)t;
              b1 = min(max(key, b0), b1);
              b0 = min(b0, key);
            }
            ++oi;
            if (last < 0 || t > last) last = t;
          }
        }
      }
      int m = -1;
      if (oi >= 2) {
        if constexpr (L2) { if (t0 >= 0 && f0 < P.nndr * f1) m = t0; }
        else if ((float)(b0 >> 16) < P.nndr * (float)(b1 >> 16)) m = (int)(b0 & 0xFFFFu);  // :744
      } else if (oi == 1) {
        m = last;                                                                    // :751-764
      }
      if (m >= 0) {
        matched[i] = m;
        atomicMin(&claim[m], i);                                                      // :776-787
      }
    }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    n_finite += __shfl_xor(n_finite, off);
    n_proj += __shfl_xor(n_proj, off);
  }
  if (lane == 0) {
    atomicAdd(&misc[0], n_finite);
    atomicAdd(&misc[1], n_proj);
  }
  __syncthreads();
  n_finite = misc[0];
  n_proj = misc[1];
  SF_TRACE_MARK(P, pair, 9);
  // id-ordered compaction
  int running = 0;
  for (int base = 0; base < Kf; base += NT) {
    const int i = base + tid;
    int m = (i < Kf) ? matched[i] : -1;
    const bool flag = (m >= 0) && (claim[m] == i);
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
    if (flag) out[running + woff + before] = (uint32_t)i | ((uint32_t)m << 16);
    running += total;
    __syncthreads();
  }
  const int n_corr = running;

  const bool outside = (n_proj == 0);                                   // :820-823
  const int words_from = outside ? 0 : n_finite;
  const int words_to_2d = outside ? 0 : Kt;
  const int words_to = (outside || mT.y <= 0) ? 0 : Kt;
  const bool pnp = P.estimation_type == 1;   // guided matching implies a calibrated camera
  // (PnP with Vis/ForwardEstOnly = false: either direction's gate, and every pair with a correspondence goes on -- as
  //  in the global matcher, k_match.hip)
  const bool both = pnp && P.bidirectional;
  const bool motion = words_to_2d > 0 &&
                      ((words_from >= P.min_inliers && (pnp ? words_to_2d : words_to) >= P.min_inliers) ||    // :1117-1118 / :1070-1071
                       (both && words_to >= P.min_inliers && words_from >= P.min_inliers));
  const bool survivor = both ? (motion && n_corr > 0) : (motion && n_corr >= P.min_inliers && n_corr >= (pnp ? 4 : 3));
  if (motion && !survivor) {
    const float* xT = st.xyz + (size_t)sT * kcap * 3;
    for (int i = tid; i < n_corr; i += NT) {
      uint32_t c = out[i];
      const float* a = xF + 3 * (c & 0xFFFFu);
      const float* b = xT + 3 * (c >> 16);
      bool ok = sfd::finite3(a[0], a[1], a[2]);
      if (!pnp)
        ok = ok && sfd::finite3(b[0], b[1], b[2]) && (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f) &&
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
    h.words_to_2d = words_to_2d;
    hdr_out = h;
    guided_flag_out = 1;
    PassState ps;
#pragma unroll
    for (int i = 0; i < 12; ++i) ps.T[i] = 0.f;
    ps.var = 1.0; ps.var_ang = 1.0;
    ps.is_null = 1;
    ps.inliers = 0;
    ps.matches = (motion && !survivor) ? misc[2] : 0;
    ps.pad = 0;
    pass2_out = ps;
    if (survivor && list) {
      int pos = atomicAdd(counter, 1);
      list[pos] = pair;
    }
  }
  SF_TRACE_MARK(P, pair, 10);
  return survivor;
}

template <int W, bool L2 = false>
__global__ void __launch_bounds__(SF_BLOCK)
k_guided(StoreView st, const int32_t* __restrict__ pair_from, const int32_t* __restrict__ pair_to,
         const PassState* __restrict__ pass1, PassState* __restrict__ pass2, uint8_t* __restrict__ guided_flag,
         uint32_t* __restrict__ corr, CorrHeader* __restrict__ hdr, int32_t* __restrict__ list,
         int32_t* __restrict__ counter, DeviceParams P) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  const int pair = (int)blockIdx.x;
  guided_body<W, L2>(st, pair, pair_from[pair], pair_to[pair], pass1[pair], pass2[pair], guided_flag[pair],
                     corr + (size_t)pair * st.kcap, hdr[pair], list, counter, P, smem);
}

// ---- result assembly: myRegistration.cpp:279-295 covariance clamp + MsgConversion.cpp:61-81 ------
__device__ __forceinline__ void finalize_one(const PassState& pass1, const PassState& pass2, uint8_t guided_flag,
                                             sf_result& out) {
  const PassState a = pass1, b = pass2;
  sf_result r;
#pragma unroll
  for (int k = 0; k < 3; ++k) r.position[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.orientation[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 36; ++k) r.covariance[k] = 0.0;
  double cd = b.var, ca = b.var_ang;
  if (cd <= 1e-9) cd = 1e-9;
  if (ca <= 1e-9) ca = 1e-9;
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.covariance[7 * k] = cd; r.covariance[7 * (k + 3)] = ca; }
  r.inliers = b.inliers;
  r.matches = b.matches;
  r.inliers_pass1 = a.inliers;
  r.matches_pass1 = a.matches;
  r.success = b.is_null ? 0 : 1;
  r.pass1_success = a.is_null ? 0 : 1;
  r.pass2_guided = guided_flag;
#pragma unroll
  for (int k = 0; k < 5; ++k) r.pad[k] = 0;
  if (!b.is_null) {
    // Eigen rotation-matrix -> quaternion, then tf::poseEigenToMsg's w >= 0 convention
    double m[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = 0; q < 3; ++q) m[p][q] = (double)b.T[4 * p + q];
      r.position[p] = (double)b.T[4 * p + 3];
    }
    double x, y, z, w;
    const double tr = (m[0][0] + m[1][1]) + m[2][2];
    if (tr > 0.0) {
      double t = sqrt(tr + 1.0);
      w = 0.5 * t;
      t = 0.5 / t;
      x = (m[2][1] - m[1][2]) * t;
      y = (m[0][2] - m[2][0]) * t;
      z = (m[1][0] - m[0][1]) * t;
    } else if (m[0][0] >= m[1][1] && m[0][0] >= m[2][2]) {   // i = 0
      double t = sqrt(((m[0][0] - m[1][1]) - m[2][2]) + 1.0);
      x = 0.5 * t;
      t = 0.5 / t;
      w = (m[2][1] - m[1][2]) * t;
      y = (m[1][0] + m[0][1]) * t;
      z = (m[2][0] + m[0][2]) * t;
    } else if (m[1][1] > m[0][0] && m[1][1] >= m[2][2]) {    // i = 1
      double t = sqrt(((m[1][1] - m[2][2]) - m[0][0]) + 1.0);
      y = 0.5 * t;
      t = 0.5 / t;
      w = (m[0][2] - m[2][0]) * t;
      z = (m[2][1] + m[1][2]) * t;
      x = (m[0][1] + m[1][0]) * t;
    } else {                                                  // i = 2
      double t = sqrt(((m[2][2] - m[0][0]) - m[1][1]) + 1.0);
      z = 0.5 * t;
      t = 0.5 / t;
      w = (m[1][0] - m[0][1]) * t;
      x = (m[0][2] + m[2][0]) * t;
      y = (m[1][2] + m[2][1]) * t;
    }
    if (w < 0.0) { x = -x; y = -y; z = -z; w = -w; }
    r.orientation[0] = x; r.orientation[1] = y; r.orientation[2] = z; r.orientation[3] = w;
  }
  out = r;
}

__global__ void __launch_bounds__(SF_BLOCK)
k_finalize(int n, const PassState* __restrict__ pass1, const PassState* __restrict__ pass2,
           const uint8_t* __restrict__ guided_flag, sf_result* __restrict__ out) {
  const int i = blockIdx.x * SF_BLOCK + threadIdx.x;
  if (i >= n) return;
  finalize_one(pass1[i], pass2[i], guided_flag[i], out[i]);
}

}  // namespace

size_t sf_guided_lds_bytes(int kcap, int n_cells, bool narrow) {
  // claim, matched, misc, cell_start, cell_fill, (+3: 16-byte alignment of the item block) item4, then the search's
  // projections (2), candidate summaries, best / second-best keys (3) and the combination list
  // claim, matched, misc, cell_start (the fill counters live in `matched` when they fit), alignment, item4, candidate
  // summaries, best / second-best keys and the combination list
  return (size_t)(2 * kcap + 16 + n_cells + 1 + (n_cells <= 2 * kcap ? 0 : n_cells) + 3 + 4 * kcap + 3 * kcap +
                  (narrow ? GUIDED_CAND_CAP_NARROW : GUIDED_CAND_CAP) +
                  (kcap > (narrow ? GUIDED_CAND_CAP_NARROW : GUIDED_CAND_CAP) ? kcap - (narrow ? GUIDED_CAND_CAP_NARROW : GUIDED_CAND_CAP) : 0)) *
         sizeof(int);   // (+ the raw keypoints' overhang: they are staged over oilast / key1 / key2 / cand, 4 kcap words)
}

int sf_launch_guided(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n) {
  if (n <= 0) return SF_OK;
  const int nc = c->dparams.grid_gx * c->dparams.grid_gy;
  const size_t lds = sf_guided_lds_bytes(st.kcap, nc);
  int32_t* counters = (int32_t*)c->counters.p;
  sf_prof_begin(c, SF_K_GUIDED);
  if (c->params.desc_type == 1) {
    if (st.w == 64)
      hipLaunchKernelGGL((k_guided<64, true>), dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                         (const PassState*)c->pass1.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                         (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (int32_t*)c->list3.p, counters + 2, c->dparams);
    else
      hipLaunchKernelGGL((k_guided<128, true>), dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                         (const PassState*)c->pass1.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                         (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (int32_t*)c->list3.p, counters + 2, c->dparams);
  } else if (st.w == 8) {
    hipLaunchKernelGGL(k_guided<8>, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                       (const PassState*)c->pass1.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                       (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (int32_t*)c->list3.p, counters + 2,
                       c->dparams);
  } else {
    hipLaunchKernelGGL(k_guided<16>, dim3(n), dim3(SF_BLOCK), lds, c->stream, st, d_from, d_to,
                       (const PassState*)c->pass1.p, (PassState*)c->pass2.p, (uint8_t*)c->flags.p,
                       (uint32_t*)c->corr2.p, (CorrHeader*)c->hdr2.p, (int32_t*)c->list3.p, counters + 2,
                       c->dparams);
  }
  sf_prof_end(c, SF_K_GUIDED);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

int sf_launch_finalize(sf_context* c, int n, sf_result* d_out) {
  if (n <= 0) return SF_OK;
  hipLaunchKernelGGL(k_finalize, dim3((n + SF_BLOCK - 1) / SF_BLOCK), dim3(SF_BLOCK), 0, c->stream, n,
                     (const PassState*)c->pass1.p, (const PassState*)c->pass2.p, (const uint8_t*)c->flags.p, d_out);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}
