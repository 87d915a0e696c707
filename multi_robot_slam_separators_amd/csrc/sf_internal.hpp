// sf_internal.hpp -- host-side context and device views shared by the kernel translation units.
// Product code (gfx950 only). No CPU fallback exists anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sepfinder.h"
#include "../../include/sf_experimental.h"

#define SF_BLOCK 256            // every verification kernel runs 256-thread workgroups (4 waves)
#define SF_MAX_KCAP 4096        // per-keyframe feature capacity ceiling of the GPU kernels
#define SF_STEP_MAX_DEPTH 16     // sf_step_issue: steps in flight at most (SF_OPT_STEP_DEPTH)
#define SF_STEP_MAX_LANES 4      // ... dealt over at most this many streams (SF_OPT_STEP_LANES)

// ---- device-resident keyframe store (one arena per field, fixed per-slot stride) ---------------
// desc : [slots][kcap][W] uint32   W = 8 (<=256-bit descriptors) or 16 (<=512-bit), zero padded
// xyz  : [slots][kcap][3] float    wire layout of KeyPoint3DVec (base frame)
// kp   : [slots][kcap]    float4   {pt.x, pt.y, bit-cast sign-extended (octave & 255), 0}
// meta : [slots]          int4     {rows, n3d, nkp, cols}
struct StoreView {
  const uint32_t* desc;
  const float* xyz;
  const float4* kp;
  const int4* meta;
  int kcap;   // features per slot (multiple of 64)
  int w;      // dwords per descriptor
  int n_slots;  // valid slots (device-side guard for caller-provided slot indices)
};

// Outcome of one registration pass for one pair (RegistrationVis result + RegistrationInfo)
struct PassState {
  float T[12];      // row-major 3x4, p_from = T p_to ; all zero when null
  double var;       // covariance diagonal before the 1e-9 clamp: linear block (all six for 3D-3D)
  double var_ang;   // angular block (PnP estimator; equals var for 3D-3D)
  int32_t is_null;
  int32_t inliers;
  int32_t matches;
  int32_t pad;
};

// Header written by the matching kernels in front of each pair's correspondence list
struct CorrHeader {
  int32_t n_corr;        // id-aligned correspondences (ascending "from" index)
  int32_t words_from;    // words3From.size()
  int32_t words_to;      // words3To.size()
  int32_t words_to_2d;   // wordsTo.size()
};

// Accepted results streamed out of the fused kernel (device-resident block the kernel reads at the END of a pair):
// records / index / flags are host-pinned (device-accessible) arrays, counter a device word zeroed before the launch.
struct AcceptStream {
  sf_result* records;      // [cap] accepted results in the order they were produced
  int32_t* index;          // [cap] pair index of each record
  uint8_t* flags;          // [pairs] success of every pair (may be null)
  unsigned* counter;       // device
  sf_result* records2;     // [cap] optional second copy of every record (e.g. a collective's send buffer on the device)
  int32_t cap;
  int32_t ext_counter;     // the counter is the caller's (zeroed by the caller before each query)
};

struct DeviceParams {
  float nndr;
  int32_t min_inliers;
  int32_t iterations;
  int32_t refine_iterations;
  double refine_sigma;
  double inlier_thr;       // (double)inlier_distance
  int32_t adaptive_stop;
  int32_t max_sample_checks;
  uint64_t seed;
  int32_t guess_win;
  int32_t calibrated;
  double fx, fy, cx, cy;
  float wlim, hlim;        // image_width-1, image_height-1
  float L[12];             // local transform
  int32_t dbg_stop;        // diagnostics only: truncate k_ransac after phase N (0 = full kernel)
  int32_t grid_gx, grid_gy;   // guided matching: uniform grid over the image (cell >= window radius)
  float grid_inv_cell;
  int32_t estimation_type;    // 0 = 3D->3D, 1 = PnP
  float pnp_thr2f;            // (float)(pnp_reproj_error^2)
  float pnp_reproj_error;
  int32_t pnp_refine_iterations;
  int32_t bundle_adjustment;  // two-view BA after each pass's estimate (k_ba.hip)
  int32_t ba_iterations;
  float ba_robust_kernel_delta, ba_pixel_variance, stereo_baseline;
  int32_t force_3dof;         // Reg/Force3DoF
  int32_t bidirectional;      // Vis/ForwardEstOnly = false (stage pipeline only)
  int32_t dbg_corr;           // fused kernel: also copy correspondence lists / headers / pass states to the global
                              // workspace (SF_OPT_DEBUG_CORR; sf_debug_correspondences)
  int32_t accept_on;          // chain kernels: accepted results also stream to the host as they are produced, into ...
  AcceptStream accept;        // ... this block (by value in the kernel arguments: nothing to upload, nothing to keep in step)
  unsigned long long* dbg_trace;   // SF_CHAIN_TRACE builds only (tools/chain_trace.py): [pair][SF_TRACE_SLOTS] timestamps; else null
};

// Phase timestamps of the fused kernel's chain (diagnostic build -DSF_CHAIN_TRACE, libsepfinder_trace.so; the product
// build compiles these to nothing): thread 0 of the pair's workgroup stores the 100 MHz wall clock.
#ifdef SF_CHAIN_TRACE
#define SF_TRACE_SLOTS 48
#define SF_TRACE_MARK(P, pair, slot)                                                              \
  do {                                                                                            \
    if (threadIdx.x == 0 && (P).dbg_trace) (P).dbg_trace[(size_t)(pair) * SF_TRACE_SLOTS + (slot)] = wall_clock64(); \
  } while (0)
// (the same with the pair's row pointer, for bodies that do not see the parameter block)
#define SF_TRACE_ROW_MARK(row, slot) do { if (threadIdx.x == 0 && (row)) (row)[slot] = wall_clock64(); } while (0)
#else
#define SF_TRACE_SLOTS 48
#define SF_TRACE_MARK(P, pair, slot) do { } while (0)
#define SF_TRACE_ROW_MARK(row, slot) do { } while (0)
#endif

// Where the fused verification kernel takes its pairs from when they are NN candidates still on the device
// (sf_find_matches_and_verify_device): candidate i = (local row, received column) -> pair (slot_other + column,
// slot_local + row); slots past *count, and candidates outside the databases / the store, are void pairs (-1, -1).
// cand == nullptr: the pair_from / pair_to arrays are used.
struct PairSource {
  const uint2* cand = nullptr;
  const unsigned* count = nullptr;
  int n_l = 0, n_r = 0, slot_other = 0, slot_local = 0, n_slots = 0;
};

struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Store {
  Buf desc, xyz, kp, meta;
  int kcap = 0, w = 0, slots = 0, cap_slots = 0;
};

struct NNDb {
  Buf rows;        // float32 [cap][dim]
  Buf norms;       // float32 [cap]
  Buf rows_h;      // fp16 [cap][dim] (nn_precision == 1)
  int n = 0, cap = 0;
  int ld = 0;      // row pitch (floats) `rows` was allocated and zero-padded for; a change re-allocates (nn_reserve)
  Buf norms_k;     // float32 [cap] squared norm of the first h_kprefix elements (filter stage)
  int h_n = -1, h_ld = 0, h_kprefix = 0;   // rows / pitch / prefix covered by the fp16 copy
  float h_scale = 1.f;      // power-of-two scale applied before the fp16 conversion
};

struct ProfSlot {
  int64_t launches = 0;
  double total_ms = 0.0;
};

struct sf_context {
  sf_params params;
  DeviceParams dparams;
  int device = 0;
  int n_cus = 0;            // compute units of the device (queried on first use)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  Store store;        // persistent keyframes
  Store scratch;      // staging slots for host-buffer calls
  Buf stage_desc, stage_xyz, stage_kp;   // H2D bounce buffers of the host-buffer ingest path
  // host-buffer batch ingest (store_add_host_batch): pinned staging, table scratch and the packing workers all
  // belong to the handle and end with it (sf_destroy)
  void* ingest_pinned = nullptr;
  size_t ingest_pinned_bytes = 0;
  struct IngestPool* ingest_pool = nullptr;
  // per-tick NetVLAD append from host float64 rows (sf_nn_append): grow-only pinned + device staging, re-used once
  // the previous append's copy has left it (nn_stage_done), so the path neither allocates nor synchronises
  void* nn_stage_pinned = nullptr;
  size_t nn_stage_pinned_bytes = 0;
  Buf nn_stage_dev;
  hipEvent_t nn_stage_done = nullptr;
  bool nn_stage_busy = false;

  // verification workspace (sized for `ws_pairs` pairs)
  int ws_pairs = 0, ws_kcap = 0;
  Buf pair_from, pair_to;       // int32[n]
  Buf corr1, corr2;             // uint32[n][kcap]
  Buf hdr1, hdr2;               // CorrHeader[n]
  Buf pass1, pass2;             // PassState[n]
  Buf ft_counts;                // batched feature extraction: corners per image (device)
  Buf pass_back, dir_mask;      // Vis/ForwardEstOnly = false: the backward estimate's PassState[n], inlier masks [2][n][kcap]
  Buf list1, list3;             // int32[n] work lists (RANSAC pass 1, RANSAC pass 2)
  Buf counters;                 // int32[8]
  Buf results;                  // sf_result[n]
  Buf flags;                    // uint8[n] pass2_guided
  // feature extraction (k_extract.hip): integral image, per-corner scratch, the BRIEF test table
  Buf ex_integral, ex_desc, ex_xyz, ex_keep, ex_rows, brief_tests;
  // corner detection (k_gftt.hip): derivative-product / response planes, candidate keys (in + sorted), sort scratch,
  // the selection's cell lists, three scalars
  Buf gf_planes, gf_keys, gf_tmp, gf_lists, gf_scalar;
  Buf lk_pyr;                      // pyramid levels >= 1 of both images (k_lk.hip)
  Buf ft_images, ft_kpts, ft_flow, ft_wire;   // sf_get_features_and_descriptor: device copies of the pair, corners, LK outputs, response
  struct sf_netvlad_model* netvlad = nullptr;   // NetVLAD inference (k_cnn.hip): weights + activation buffers
  int brief_bytes = 0;                 // 0: table not uploaded yet
  int8_t brief_host[64 * 8 * 4] = {};
  Buf trace;                    // SF_CHAIN_TRACE builds: uint64[n][32] phase timestamps of the fused kernel

  // NN stage
  NNDb nn_local, nn_recv;
  int nn_dim = 0;
  std::vector<uint8_t> mask_local, mask_other;         // host mirrors
  std::vector<int32_t> ignored;                        // pairs (local, other)
  Buf d_mask_local, d_mask_other, d_ign_ptr, d_ign_col;
  bool masks_dirty = true;
  Buf nn_rowmin;     // per-row packed (dist bits, idx) uint64 [n_local]
  Buf nn_exact;      // double [n_local]
  Buf nn_cand;       // filter path: two 64-byte counter blocks (alternating per launch) + candidate (row, col) pairs + exact distances
  int nn_count_idx = 0;          // counter block of the next filter launch
  bool nn_count_primed = false;  // ... and whether the previous k128 launch already zeroed it
  Buf nn_scalar;     // small reduction scratch
  int nn_level = 0, nn_level_cooldown = 32, nn_last_kdims = 0;   // adaptive prefix ladder of the filter
  bool nn_force_full = false;   // SF_OPT_NN_FULL_FILTER: always contract the full descriptor length
  int nn_coef_level = -1, nn_coef_nl = 0, nn_coef_nr = 0;         // what the cached filter coefficients were built for
  double nn_coef_thr = 0.0;
  float nn_coef_scale = 0.f;
  const void* nn_coef_ptr = nullptr;
  std::vector<double> last_row_min;
  std::vector<int32_t> last_row_arg;
  std::vector<uint64_t> nn_sort_keys, nn_sort_keys2;   // host scratch of the walk (kept to avoid reallocation)
  std::vector<int32_t> nn_sort_rows, nn_sort_rows2;
  std::vector<uint32_t> nn_sort_hist;
  std::vector<uint8_t> nn_taken;

  int match_variant = 0;   // 0 = default geometry; see sf_launch_match_global
  bool ransac_attr_set = false;
  bool pnp_attr_set = false;
  bool fused_attr[2][2][2] = {};   // [W == 16][matrix-core matcher][WIDE]: LDS attribute set
  bool ransac_ba_attr_set = false, pnp_ba_attr_set = false, merge_ba_attr_set = false;
  bool debug_corr = false;      // SF_OPT_DEBUG_CORR: the fused kernel also writes lists / headers / states to HBM
  bool last_lists_valid = false;   // the last verification left correspondence lists in the global workspace
  bool fused = true;        // fused per-pair verification kernel (SF_FUSED=0 selects the stage kernels)
  int cu_count = 0;         // compute units of the device (grids sized to the chip); filled on first use
  bool gf_select_attr = false; // k_gftt_select_lds: dynamic LDS attribute set
  bool nn_k128_attr = false;   // k_nn_filter_f16_k128: dynamic LDS attribute set
  bool split = false;       // SF_FUSED=2: one matching launch + one chain launch over the survivors (k_verify.hip)
  bool split_auto = true;   // SF_OPT_STEP_SPLIT: the split form inside overlapped steps (sf_use_split, sf_api.hip); on again since
                            // round 5 (its matcher is software-pipelined: 23.0 against 22.7 M pairs/s, profiles/r05u_*)
  int split_auto_min = 2048;   // ... for queries of at least this many candidates (SF_STEP_SPLIT_MIN): below, one launch wins
  bool in_overlapped_step = false;   // set around sf_step_issue's body while the steps alternate between two streams
  bool chain_attr[2][3][3] = {};   // k_chain [W == 16][part 0 / 1 / 2][wavefronts per chain 1 / 2 / 4]: LDS attribute set
  int chain_nw = 4;                // wavefronts per motion-estimation chain of the split forms: 1, 2 (round 5), or 4 = the
                                   // 256-thread workgroup of rounds 2-4 (SF_CHAIN_NW)
  bool split_match_attr[2] = {};   // k_match_split [W == 16]
  bool chain_pnp_attr[2][3][3] = {};  // k_chain_pnp [W == 16][part 0 / 1 / 2][wavefronts 1 / 2 / 4]
  int chain_pnp_nw = 2;            // wavefronts per PnP chain (SF_CHAIN_PNP_NW; 2 measured fastest: profiles/r05i_pnp_chain_width.txt)
  bool ba_pass_attr[2][3] = {};    // k_ba_pass [PnP][wavefronts 1 / 2 / 4]
  int ba_occ = 0;                  // wavefronts per SIMD the SMALL adjustment kernel is compiled for (SF_BA_OCC; 1 = 512 registers,
                                   // no scratch; 2 = 256 registers + 256 B of scratch; 0 = by estimator: PnP 1, 3D-3D 2 -- measured,
                                   // profiles/r05e_ba_occupancy.txt)
  int ba_nw = 1;                   // wavefronts per bundle adjustment (k_ba_pass; SF_BA_NW): 1 measured fastest (profiles/r05c_ba_width.txt)
  bool chain_pnp = true;           // PnP estimator: k_match_split + k_chain_pnp instead of the five stage launches
                                   // (SF_CHAIN_PNP=0: the stage launches)
  bool match_mfma = true;   // Hamming table on the fp4 matrix cores (SF_MATCH_MFMA=0 selects the VALU matcher)
  void* nn_pinned = nullptr;   // pinned host staging of the NN filter's small D2H copies
  size_t nn_pinned_bytes = 0;

  // sf_verify_matches_device / sf_compact_accepted_device staging
  void* pairs_pinned = nullptr;
  size_t pairs_pinned_bytes = 0;
  hipEvent_t pairs_staged = nullptr;
  int32_t* count_pinned = nullptr;
  Buf compact_scratch;          // per-chunk counts of sf_compact_accepted_device

  // multi-GPU exchange (sf_comm.hip)
  void* comm = nullptr;    // ncclComm_t
  int comm_rank = 0, comm_world = 1;
  int32_t comm_count_host = 0;
  Buf comm_scratch;

  // Two-stream verification of large batches (sf_api.hip, verify_device): the second half of a batch runs
  // its stage kernels on `twin` -- a shadow context with its own stream, workspace and counters that shares
  // this handle's keyframe store and parameters -- so that the latency-bound motion-estimation kernels of
  // one half overlap the issue-bound matching kernel of the other.  Off unless SF_OVERLAP=1 (+3-6 %).
  sf_context* twin = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = false;
  int overlap_min_pairs = 4096;
  int ws_split = 0;             // pairs [0, ws_split) of the last batch live in this workspace, the rest in twin's

  // Speculative verification (sf_find_matches_and_verify_device): every candidate (row, column) the NN filter
  // emits is verified on the device while the host still reduces the candidates to row minima, sorts and
  // walks them; the walk's matches then pick their results out of the speculative ones.
  // results of the last sf_find_matches_and_verify_device: record of match i = last_results[index ? index[i] : i]
  const sf_result* last_results = nullptr;
  const int32_t* last_results_index = nullptr;
  int last_results_n = 0;
  unsigned compact_epoch = 0;   // k_compact_chain: tag of the current launch's prefix entries
  int compact_state_chunks = 0; // ... and how many state entries have been initialised
  void* compact_state_ptr = nullptr;
  PairSource pair_src;          // candidate list the fused kernel derives its pairs from (speculative path), or empty
  struct Spec {
    bool requested = false;     // set by the entry point for the duration of one sf_nn_run
    bool launched = false;      // the candidates of some filter level were handed to the verification
    bool valid = false;         // ... and that level's candidate set was accepted (n_cand <= grid)
    int32_t slot_other = 0, slot_local = 0;
    unsigned grid = 0;          // speculative pair slots (candidates beyond it invalidate the speculation)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_refined = nullptr, ev_copied = nullptr;
  } spec;
  // accepted-result streams (sf_accept_stream_set / _select): two registered blocks, the one selected for the next
  // speculative query is handed to the fused kernel; `streamed` says whether the last query used it
  struct AcceptHost { AcceptStream s = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}; bool set = false; } accept_blocks[3];   // [2]: the step pair's own
  int accept_sel = -1;
  bool accept_streamed = false;     // the last query's results are in the selected block (keyed by verified slot)
  bool accept_armed = false;        // the selected block was handed to a verification launch of the last query (it may
                                    // hold records even when the query then fell back: streamed = false)
  Buf spec_from, spec_to, spec_results, spec_index;
  // sf_step_issue / sf_step_retire: the caller's loop body (find_separators.py:59-133) as a begin / retire pair.  A ring
  // of `step_depth + 1` blocks: up to `step_depth` steps in flight, and the block of the step retired last stays
  // untouched until the next retire.
  struct StepBlock {
    void* pinned = nullptr;            // ONE host-pinned allocation: records | index | flags | count | matches | walk words
    size_t pinned_bytes = 0;
    sf_result* records = nullptr;      // [cap] accepted results (streamed: completion order; compacted: match order)
    int32_t* index = nullptr;          // [cap] streamed: verified slot of each record, -1 = unused entry
    uint8_t* flags = nullptr;          // [cap] compacted: success of every match
    int32_t* count = nullptr;          // compacted: number of accepted matches
    sf_match* walk_matches = nullptr;  // [cap] device walk: the matches, written by k_walk_emit
    int32_t* walk_slots = nullptr;     // [cap] speculative device step: the verification slot (= candidate index) of each match
    int32_t* walk_n = nullptr;         // device walk: their number ...
    int32_t* walk_status = nullptr;    // ... and the NN stage's status (0, or 1 = candidate set denser than the filter level allows)
    int32_t cap = 0;
    Buf dev;                           // device: 64-byte counter block {matches, -, -, -, accept slot counter} + (row, column)[cap]
    Buf dev_records;                   // device: [cap] the accepted records once more, same slots as `records` (sf_step_result.d_records)
    std::vector<sf_match> matches;     // host walk: the matches
    std::vector<int32_t> slot_of_match, record_of_match, rec_of_slot;
    int32_t n = 0, pairs = 0;
    bool streamed = false, armed = false, issued = false;
    bool speculative = false;          // device walk, speculative form: record of match i = the record of slot walk_slots[i]
    bool device_walk = false;          // the NN walk ran on the device: matches / n come from the pinned block at retire
    bool settled = false;              // its `done` event has been waited for (and a fallback, if needed, has run)
    int32_t slot_other = 0, slot_local = 0, parity = 0;   // (what a fallback needs to run the query again)
    int settle_rc = 0;
    hipEvent_t done = nullptr;
    // a caller's copy out of dev_records (sf_memcpy_device_async, on ANY stream) is recorded here; the step that next
    // uses the block waits for it before its kernels may write the buffer again
    hipEvent_t copied = nullptr;
    bool copy_pending = false;
  } step_blocks[SF_STEP_MAX_DEPTH + 1];
  int step_depth = 6;                      // SF_OPT_STEP_DEPTH: steps in flight
  int step_lanes = 3;                      // SF_OPT_STEP_LANES: streams the steps in flight are dealt over (step k on lane k mod lanes)
  bool step_device_walk = true;            // SF_OPT_STEP_DEVICE_WALK: no host wait inside sf_step_issue
  uint64_t step_seq = 0;                   // steps issued so far (the next step's number)
  int step_inflight = 0;
  // SF_OPT_STEP_OVERLAP: the steps in flight run on `step_lanes` streams -- lane k > 0 with its own copy of every device
  // buffer a step writes (parked in `lanes[k - 1]` while another lane's step owns the handle's members of the same names,
  // swapped in for the duration of sf_step_issue) -- so that the tail of one step's verification (its last
  // motion-estimation chains on an emptying chip) and the NN stage of the next overlap.
  struct StepLane {
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;             // speculative device step: exact re-evaluation + minima + walk beside the verification
    hipEvent_t ev_filter = nullptr, ev_walk = nullptr;
    hipEvent_t ev_main = nullptr;          // "the handle's stream up to here": awaited when the databases changed
    Buf pair_from, pair_to, corr1, corr2, hdr1, hdr2, pass1, pass2, pass_back, dir_mask, list1, list3, counters, results,
        flags, nn_cand, spec_from, spec_to, spec_results, spec_index, compact_scratch, step_nn, walk_scratch;
    int ws_pairs = 0, ws_kcap = 0, nn_count_idx = 0;
    bool nn_count_primed = false;
    unsigned compact_epoch = 0;
    int compact_state_chunks = 0;
    void* compact_state_ptr = nullptr;
    uint64_t seen_db_epoch = ~0ull;
  } lanes[SF_STEP_MAX_LANES - 1];
  // Where the pipeline's streams sit on the hardware (sf_api.hip: place_streams).  Measured once, at the first step that
  // needs a second stream: a launch that does not fit on the chip keeps the dispatcher of its queue's PIPE busy until
  // its last workgroup is placed, and every other queue of that pipe waits -- so the lanes' main streams are picked
  // from candidates on different pipes and the second streams (short chains of small launches) from a pipe none of
  // them uses.  Streams are handed out once; what is left over is destroyed with the handle.
  struct StreamPlacement {
    bool tried = false, done = false;
    hipStream_t main[SF_STEP_MAX_LANES] = {};   // [0] unused: lane 0 runs on the handle's stream
    hipStream_t aux[SF_STEP_MAX_LANES] = {};
    hipStream_t copy = nullptr;                  // the second stream of the synchronous speculative call
    char report[384] = {0};
  } placement;
  int cur_lane = 0;                        // the lane sf_step_issue is issuing on
  Buf step_nn;                             // device walk: row minima (f64) | row arg (i32) | row candidate (i32) | packed arg (u64) | status
  hipStream_t aux = nullptr;               // (lane 0's; swapped with the lanes' like `stream`)
  hipEvent_t ev_filter = nullptr, ev_walk = nullptr;
  bool step_speculate = true;              // SF_OPT_STEP_SPECULATE: batch-mode steps verify every filter candidate beside the walk
  Buf walk_scratch;                        // device walk: tile keys / rows, sorted rows, column claims (sf_nn_walk_dev)
  // Work that prepares state ALL lanes read (fp16 copies of the databases, filter coefficients, masks) is queued by
  // whichever lane first needs it; it bumps prep_count, the issuing step records ev_prep behind it and the other lanes
  // wait for that event once (lane_seen_prep).
  uint64_t prep_count = 0, prep_epoch = 0;
  uint64_t lane_seen_prep[SF_STEP_MAX_LANES] = {0, 0, 0, 0};
  hipEvent_t ev_prep = nullptr;
  bool step_overlap = true;                // the option (SF_STEP_OVERLAP=0 / sf_set_option turn it off)
  uint64_t db_epoch = 0;                   // bumped by every call that writes a database through the handle's stream
  sf_result* step_mirror_records[2] = {nullptr, nullptr};   // sf_step_mirror[_pair]: second (device) destination of every
  uint32_t* step_mirror_counter[2] = {nullptr, nullptr};    // accepted record and the caller's slot counter, per step parity
  int32_t step_mirror_cap = 0;
  bool step_mirror_lanes = false;             // sf_step_mirror_streams: the odd steps of a mirror pair run on lane 1
  void* spec_index_pinned = nullptr;
  size_t spec_index_pinned_bytes = 0;
  hipEvent_t spec_index_staged = nullptr;
  std::vector<int32_t> last_row_cand;   // candidate-list index of each row's minimum (filter path)

  // profiling
  bool prof = false;
  uint32_t prof_mask = 0xFFFFFFFFu;      // bit k: kernel k is bracketed when profiling is on (sf_prof_select)
  std::vector<hipEvent_t> prof_event_pool;   // timing events are reused, not created per launch
  ProfSlot prof_slots[SF_K_COUNT];
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
};

// ---- helpers implemented in sf_api.hip ---------------------------------------------------------
int sf_fail(sf_context* c, int code, const char* fmt, ...);
// a database is about to change (SF_OPT_STEP_OVERLAP): `drain` also waits for the step in flight on the second stream
int sf_lanes_touch(sf_context* c, bool drain);
int sf_buf_reserve(sf_context* c, Buf& b, size_t bytes, bool keep = false);
void sf_buf_free(Buf& b);
struct sf_netvlad_model;
void sf_netvlad_free(sf_context* c);
int sf_netvlad_load_impl(sf_context* c, const sf_netvlad_weights* w);
int sf_netvlad_infer_impl(sf_context* c, const float* d_image, int H, int W, float* d_out, int n_out);
int sf_netvlad_infer_batch_impl(sf_context* c, const float* d_images, int n_img, int H, int W, float* d_out, int n_out);
StoreView sf_store_view(const Store& s);
void sf_prof_begin(sf_context* c, int kernel);
void sf_prof_end(sf_context* c, int kernel);

// gate selector of the matching kernels: 0 = 3D->3D, 1 = PnP, 2 = PnP without a calibrated camera, 3 = PnP with
// Vis/ForwardEstOnly = false (either direction's gate sends the pair on; each estimate then checks its own)
inline int sf_est_mode(const sf_context* c) {
  return c->dparams.estimation_type == 1 ? (c->dparams.calibrated ? (c->dparams.bidirectional ? 3 : 1) : 2) : 0;
}

// Gate of one direction of the PnP estimate (myRegistrationVis.cpp:1059, :1070-1071 with A / B as :936-977 set them):
// dir 0: 3D words of "from" and 2D words of "to"; dir 1: 3D words of "to" and 2D words of "from" -- every "from" row
// after global matching (:856-875), the rows with a finite point after guided matching (:766-774).
__host__ __device__ inline bool sf_pnp_dir_gate(int dir, const CorrHeader& h, int rows_from, bool guided, int min_inliers) {
  if (h.words_to_2d <= 0) return false;                                   // :928
  if (dir == 0) return h.words_from >= min_inliers && h.words_to_2d >= min_inliers;
  return h.words_to >= min_inliers && (guided ? h.words_from : rows_from) >= min_inliers;
}

#define SF_HIP(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return sf_fail((c), SF_EHIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// ---- kernel launchers (one per translation unit) -----------------------------------------------
// Pass-1 global matching for n pairs; also writes PassState defaults and the RANSAC work list.
int sf_launch_match_global(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n);
// RANSAC for the pairs in work list `list` (count in counters[ctr]); pass = 1 or 2.
int sf_launch_ransac(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass);
// PnP (estimation_type = 1) for the pairs in the same work lists.
int sf_launch_pnp(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass);
// Guided matching for pairs whose pass 1 succeeded; builds the pass-2 RANSAC work list.
int sf_launch_guided(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n);
// Vis/ForwardEstOnly = false with bundle adjustment: merge of a pass's two estimates + adjustment over the union (k_ba.hip)
int sf_launch_ba_pass(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass, int list_sel,
                      const uint8_t* mask, const uint8_t* run, bool fin, sf_result* d_out);
int sf_launch_merge_directions_ba(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n, int pass,
                                  bool pnp, const uint8_t* mask_f, const uint8_t* mask_b);
// Fused per-pair pipeline (k_verify.hip): match -> RANSAC -> guided -> RANSAC -> result in one launch.
size_t sf_fused_lds_bytes(const sf_context* c, const StoreView& st);
bool sf_split_applicable(const sf_context* c, const StoreView& st);
bool sf_split_pnp_applicable(const sf_context* c, const StoreView& st);
int sf_launch_verify_split(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n,
                           sf_result* d_out);
int sf_launch_verify_fused(sf_context* c, StoreView st, const int32_t* d_from, const int32_t* d_to, int n,
                           sf_result* d_out);
size_t sf_ransac_lds_bytes(int kcap, int iterations);
size_t sf_ba_lds_bytes(int kcap);
size_t sf_pnp_lds_bytes(int kcap, int iterations);
size_t sf_guided_lds_bytes(int kcap, int n_cells, bool narrow = false);   // narrow: the one- / two-wavefront chains' shorter candidate list
void sf_brief_default_pattern(int8_t* tests, int bytes);
int sf_launch_detect_corners_batch(sf_context* c, const uint8_t* d_images, size_t img_stride, int n_img, int width, int height,
                                   int pitch, int max_corners, double quality_level, double min_distance,
                                   sf_keypoint* d_kpts_out, int cap, int32_t* d_n_out);
int sf_launch_stereo_flow_batch(sf_context* c, const uint8_t* d_left, const uint8_t* d_right, size_t img_stride, int n_img,
                                int width, int height, int pitch, const sf_keypoint* d_kpts, int n, const int32_t* d_n,
                                const sf_stereo_flow_params* prm, float* d_right_xy, uint8_t* d_status, float* d_right_x,
                                float* d_err);
int sf_launch_extract_batch(sf_context* c, const uint8_t* d_left, size_t img_stride, int n_img, int width, int height,
                            int pitch, const sf_keypoint* d_kpts, const float* d_right_x, const uint8_t* d_status, int n,
                            const int32_t* d_n, const sf_stereo_camera* cam, int bytes, const int8_t* d_tests,
                            uint32_t* st_desc, float* st_xyz, float4* st_kp, int4* st_meta, int kcap, int w_dwords,
                            int slot, uint8_t* d_desc_out, float* d_xyz_out, sf_keypoint* d_kpts_out,
                            int32_t* d_rows_out);
int sf_launch_detect_corners(sf_context* c, const uint8_t* d_image, int width, int height, int pitch, int max_corners,
                             double quality_level, double min_distance, sf_keypoint* d_kpts_out, int cap,
                             int32_t* n_out);
int sf_launch_stereo_flow(sf_context* c, const uint8_t* d_left, const uint8_t* d_right, int width, int height, int pitch,
                          const sf_keypoint* d_kpts, int n, const sf_stereo_flow_params* prm, float* d_right_xy,
                          uint8_t* d_status, float* d_right_x, float* d_err);
int sf_launch_extract(sf_context* c, const uint8_t* d_left, int width, int height, int pitch, const sf_keypoint* d_kpts,
                      const float* d_right_x, const uint8_t* d_status, int n, const sf_stereo_camera* cam, int bytes,
                      const int8_t* d_tests, uint32_t* st_desc, float* st_xyz, float4* st_kp, int4* st_meta, int kcap,
                      int w_dwords, int slot, uint8_t* d_desc_out, float* d_xyz_out, sf_keypoint* d_kpts_out,
                      int32_t* d_rows_out);
// Assemble sf_result records.
int sf_launch_finalize(sf_context* c, int n, sf_result* d_out);
// Ingest kernels
int sf_launch_ingest(sf_context* c, Store& st, int first_slot, int n, int rows, int cols,
                     const uint8_t* d_desc, const float* d_xyz, const sf_keypoint* d_kp);
// NN stage
int sf_nn_run(sf_context* c, sf_match* out, int cap, int* n_out);
int sf_nn_row_minima_dev(sf_context* c, double* d_row_min, int32_t* d_row_arg, int32_t* d_status);
int sf_nn_walk_dev(sf_context* c, const double* d_row_min, const int32_t* d_row_arg, const int32_t* d_status, int n_l,
                   int n_r, double thr, int max_matches_nb, int cap, void* d_match_rc, unsigned* d_count_block,
                   sf_match* out_matches, int32_t* out_n, int32_t* out_status, const int32_t* d_row_cand = nullptr,
                   int32_t* out_slot = nullptr, const unsigned* d_cand_count = nullptr, unsigned cand_grid = 0);
// the filter path of sf_nn_row_minima_dev in its two halves (k_nn.hip)
struct NnFilterOut {
  const void* cand = nullptr;       // uint2 (row, column)[]
  const unsigned* count = nullptr;  // its length (device), word 4 of the same 64-byte block: the accept stream's slot counter
  double* cdist = nullptr;          // exact distances, filled by the second half
  unsigned limit = 0, typical = 0;
};
int sf_nn_filter_dev(sf_context* c, NnFilterOut* out);
int sf_nn_minima_of_candidates_dev(sf_context* c, const NnFilterOut& fo, double* d_row_min, int32_t* d_row_arg,
                                   int32_t* d_status, int32_t* d_row_cand, unsigned long long* d_arg64, bool throttle);
int sf_nn_walk_host(sf_context* c, const double* row_min, const int32_t* row_arg, int n_l, int n_r, double thr,
                    int max_matches_nb, sf_match* out, int cap, int* n_out);
// Speculation hook (sf_api.hip), called by the NN filter right behind the refinement launch of a prefix level:
// builds the candidate pair list on the device and queues the verification of every candidate.
int sf_spec_launch(sf_context* c, const void* d_cand, const unsigned* d_count);
int sf_nn_append(sf_context* c, NNDb& db, const void* src, int n, int dim, int src_kind);
