/*
 * sf_experimental.h -- entry points of libsepfinder.so that are NOT part of the drop-in surface (include/sepfinder.h).
 *
 * They are the building blocks sf_step_issue / sf_step_retire are made of, exported for the library's own tests, for
 * tools/ and for bench.py's diagnostic variants (BENCH_NO_STREAM, the N > 1 fallbacks).  No reference interface
 * corresponds to them one to one (the loop they serve is PKG/scripts/find_separators.py:59-133); a host written against
 * the reference needs none of them, and they may change without a bump of SF_ABI_VERSION.
 */
#ifndef SF_EXPERIMENTAL_H
#define SF_EXPERIMENTAL_H

#include "sepfinder.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sf_find_matches_and_verify_device with d_out = NULL leaves the results where the verification wrote them, and
   sf_last_match_results says where: the record of match i is d_results[index ? index[i] : i] (index, if not NULL, is
   device-readable pinned host memory owned by the handle, valid until the next sf_find_matches_and_verify_device).
   sf_compact_accepted_indexed_device_async consumes exactly that pair (any index == NULL means "in order"), so a
   caller that only needs the ACCEPTED separators saves the gathered copy of all of them.                        */
int  sf_last_match_results(sf_handle h, const sf_result** d_results, const int32_t** index, int32_t* n);
int  sf_compact_accepted_indexed_device_async(sf_handle h, const sf_result* d_results, const int32_t* index,
                                              int32_t n, sf_result* d_accepted, uint8_t* d_flags,
                                              int32_t* d_n_accepted);
/* The same with every output written TWICE: accepted records, flags and count also go to d_accepted2 / d_flags2
   (optional) / d_n_accepted2 -- e.g. a collective's send buffer on the device AND this rank's own copy in pinned host
   memory, without a copy behind the kernel (bench.py, N > 1).                                                     */
int  sf_compact_accepted_indexed_mirrored_device_async(sf_handle h, const sf_result* d_results, const int32_t* index,
                                                       int32_t n, sf_result* d_accepted, uint8_t* d_flags,
                                                       int32_t* d_n_accepted, sf_result* d_accepted2, uint8_t* d_flags2,
                                                       int32_t* d_n_accepted2);

/* Accepted results STREAMED out of the verification kernels (both estimators; the speculative path of
   sf_find_matches_and_verify_device): every pair whose result is accepted writes its record into the selected block
   the moment it is final -- posted writes beside the other pairs' work -- instead of a compaction kernel behind the
   launch.  A block = host-pinned (device-accessible) arrays: records [cap], index [cap] (the record's slot in the
   sf_last_match_results block, i.e. match i owns the record whose index equals index_of_match[i]), flags [pairs]
   (optional: success of EVERY verified slot, 0 for slots without a candidate -- if that array is host memory every
   pair ends on a 1-byte PCIe write its workgroup has to see acknowledged, measured as +10 us on a 10 000-pair launch:
   pass NULL and derive the flags from the index list, whose entries beyond the streamed count keep their old value).  Records arrive in completion order and
   cover every verified candidate (a superset of the matches when a row had several candidates): the host keeps those
   whose index is a match's.  Their number is the sum of flags over the `pairs` slots sf_accept_stream_status reports.
   Two blocks can be registered and selected alternately, so that one query's separators stay untouched while the
   next one runs.  `streamed` = 0 after a call that could not stream (fallback paths): use the compaction then.     */
/* d_records2 (optional): every record is written there too, at the same slot -- e.g. a collective's send buffer on the
   device.  d_counter (optional): the slot counter is this device word instead of the handle's own; the CALLER zeroes
   it before each query (e.g. the count header of that send buffer, so that the all-gather can start behind the
   verification with nothing in between).                                                                       */
int  sf_accept_stream_set(sf_handle h, int32_t which, sf_result* records, int32_t* index, uint8_t* flags, int32_t cap,
                          sf_result* d_records2, uint32_t* d_counter);
int  sf_accept_stream_select(sf_handle h, int32_t which);      /* 0 / 1, -1 = off (default) */
int  sf_accept_stream_status(sf_handle h, int32_t* streamed, int32_t* pairs);

/* Which kernels are bracketed (bit k = kernel k of the enum above; default all).  Two timing events per launch cost
   host time and a marker on the queue -- about 4 us per bracketed launch in a 0.6 ms step -- so a throughput
   measurement brackets only the kernel it prices (bench.py: the dominant one) and surveys the rest separately.  */
int  sf_prof_select(sf_handle h, uint32_t kernel_mask);
/* One line on where the step pipeline's streams were placed on the hardware's dispatch pipes (measured once per handle at
   the first step that needs a second stream: sf_api.hip place_streams; SF_STREAM_PLACEMENT=0 in the environment turns the
   measurement off, =2 also prints the line to stderr). */
int  sf_stream_placement(sf_handle h, char* buf, size_t n);
/* Runs that measurement now instead of inside the first step that needs a second stream (60-100 ms once per handle:
   ~100 short chip-filling launches on the handle's stream and on streams of the library's own; it waits for the
   handle's own streams only).  Work of other streams or processes on the GPU at that moment changes what is measured:
   call it at a quiet moment, read sf_stream_placement for what was picked or why it was abandoned.  Idempotent;
   SF_STREAM_PLACEMENT=0 makes it a no-op.                                                                        */
int  sf_streams_prepare(sf_handle h);

/* Pass state of pair `pair` of the last verification (diagnostics, tools/diag_pair_index.py): pose of the pass (row-major
   3 x 4, p_from = T p_to; all zero when null), is_null / inliers / matches.  Needs SF_OPT_DEBUG_CORR like
   sf_debug_correspondences.                                                                                     */
int  sf_debug_pass_state(sf_handle h, int32_t pair, int32_t pass, float* T12, int32_t* is_null, int32_t* inliers,
                         int32_t* matches);
/* Diagnostic counters the kernels bump when the process runs with SF_DIAG set (experiments; zero otherwise).      */
int  sf_debug_counters(sf_handle h, unsigned long long* out, int32_t n);
int  sf_debug_guided_points(sf_handle h, int32_t pair, unsigned long long* plane0, unsigned long long* plane1,
                            int32_t* kcap_out);

/* The launch plan of a verification call of n_pairs pairs on keyframes of `kcap` feature slots (a multiple of 64) and
   `desc_words` dwords per descriptor row, and its workspace -- computed on the host, no device needed (a test without a
   GPU holds the reservation against what each launch form writes: the out-of-bounds write of round 3 was a form that
   writes correspondence lists on a workspace reserved without them).  out (>= 22 values): [0] form (0 stages, 1 fused,
   2 split, 3 split PnP, 4 two-stream halves), [1] lists in HBM, [2] one chunk, [3] pairs of the largest launch
   sequence, [4..12] bytes reserved for corr1, corr2, hdr1, hdr2, pass1, pass2, list1, list3, flags, [13..21] bytes the
   form's launches write to them.                                                                                 */
int  sf_debug_plan_workspace(const sf_params* p, int32_t kcap, int32_t desc_words, int32_t n_pairs,
                             int32_t in_overlapped_step, int32_t debug_corr, int64_t* out, int32_t n_out);

#ifdef __cplusplus
}
#endif

#endif /* SF_EXPERIMENTAL_H */
