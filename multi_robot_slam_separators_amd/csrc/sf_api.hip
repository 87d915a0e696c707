// sf_api.hip -- the C-ABI of include/sepfinder.h: context, device-resident keyframe store,
// verification pipeline orchestration, measurement hooks.  gfx950 only; there is no CPU path:
// sf_create fails with SF_ENODEV when no GPU is visible.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "sf_internal.hpp"

static thread_local std::string g_create_error;

int sf_fail(sf_context* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

int sf_buf_reserve(sf_context* c, Buf& b, size_t bytes, bool keep) {
  if (bytes <= b.bytes) return SF_OK;
  size_t want = std::max(bytes, b.bytes + b.bytes / 2);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) return sf_fail(c, SF_ENOMEM, "hipMalloc(%zu) -> %s", want, hipGetErrorString(e));
  if (b.p) {
    if (keep && b.bytes) {
      e = hipMemcpyAsync(p, b.p, b.bytes, hipMemcpyDeviceToDevice, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { (void)hipFree(p); return sf_fail(c, SF_EHIP, "grow copy -> %s", hipGetErrorString(e)); }
    } else {
      (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(b.p);
  }
  b.p = p;
  b.bytes = want;
  return SF_OK;
}

void sf_buf_free(Buf& b);
static void buf_free(Buf& b) { sf_buf_free(b); }
void sf_buf_free(Buf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

StoreView sf_store_view(const Store& s) {
  StoreView v;
  v.desc = (const uint32_t*)s.desc.p;
  v.xyz = (const float*)s.xyz.p;
  v.kp = (const float4*)s.kp.p;
  v.meta = (const int4*)s.meta.p;
  v.kcap = s.kcap;
  v.w = s.w;
  v.n_slots = s.slots;
  return v;
}

// ---- profiling ------------------------------------------------------------------------------
static const char* k_names[SF_K_COUNT] = {"k_match_global", "k_ransac(pass1)", "k_guided",
                                          "k_ransac(pass2)", "k_nn_argmin", "k_nn_select",
                                          "k_nn_filter_f16", "k_nn_refine", "k_verify_fused", "k_nn_walk", "k_ba_pass"};
const char* sf_kernel_name(int k) { return (k >= 0 && k < SF_K_COUNT) ? k_names[k] : "?"; }

// Brackets that have completed are booked and their events reused without waiting for anything: a long profiled
// run then lives on a handful of events (creating two per launch made the runtime grow its signal pool in the
// middle of a timed region: one 7 ms step every few hundred launches).
static void prof_harvest(sf_context* c) {
  size_t done = 0;
  while (done < c->pending_events.size() && hipEventQuery(c->pending_events[done].second.second) == hipSuccess) {
    auto& pe = c->pending_events[done];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
      c->prof_slots[pe.first].launches += 1;
      c->prof_slots[pe.first].total_ms += (double)ms;
    }
    c->prof_event_pool.push_back(pe.second.first);
    c->prof_event_pool.push_back(pe.second.second);
    ++done;
  }
  if (done) c->pending_events.erase(c->pending_events.begin(), c->pending_events.begin() + (long)done);
}

void sf_prof_begin(sf_context* c, int kernel) {
  if (!c->prof || !((c->prof_mask >> kernel) & 1u)) return;
  if (c->prof_event_pool.size() < 2 && c->pending_events.size() >= 4) prof_harvest(c);
  hipEvent_t a = nullptr, b = nullptr;
  for (hipEvent_t* e : {&a, &b}) {
    if (!c->prof_event_pool.empty()) { *e = c->prof_event_pool.back(); c->prof_event_pool.pop_back(); }
    else if (hipEventCreate(e) != hipSuccess) { if (a) c->prof_event_pool.push_back(a); return; }
  }
  (void)hipEventRecord(a, c->stream);
  c->pending_events.push_back({kernel, {a, b}});
}

void sf_prof_end(sf_context* c, int kernel) {
  if (!c->prof || c->pending_events.empty()) return;
  auto& pe = c->pending_events.back();
  if (pe.first != kernel) return;
  (void)hipEventRecord(pe.second.second, c->stream);
}

static void prof_resolve(sf_context* c) {
  if (c->pending_events.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto& pe : c->pending_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
      c->prof_slots[pe.first].launches += 1;
      c->prof_slots[pe.first].total_ms += (double)ms;
    }
    c->prof_event_pool.push_back(pe.second.first);
    c->prof_event_pool.push_back(pe.second.second);
  }
  c->pending_events.clear();
}

// ---- ingest kernel: wire layout -> store layout ---------------------------------------------------
namespace {

// one workgroup per keyframe.  desc rows are zero padded to w dwords; keypoints are reduced to
// {x, y, sign-extended (octave & 255)} (myRegistrationVis.cpp:709-710 compares only that byte).
__global__ void __launch_bounds__(SF_BLOCK)
k_ingest(uint32_t* __restrict__ desc, float* __restrict__ xyz, float4* __restrict__ kp, int4* __restrict__ meta,
         int kcap, int w, int first_slot, int rows, int cols, int n3d, const uint8_t* __restrict__ s_desc,
         const float* __restrict__ s_xyz, const sf_keypoint* __restrict__ s_kp) {
  const int k = blockIdx.x;
  const int slot = first_slot + k;
  const int tid = threadIdx.x;
  uint8_t* d8 = reinterpret_cast<uint8_t*>(desc + (size_t)slot * kcap * w);
  const uint8_t* sd = s_desc + (size_t)k * rows * cols;
  const int rowb = w * 4;
  for (int i = tid; i < rows * rowb; i += SF_BLOCK) {
    const int r = i / rowb, b = i - r * rowb;
    d8[i] = (b < cols) ? sd[(size_t)r * cols + b] : (uint8_t)0;
  }
  float* dx = xyz + (size_t)slot * kcap * 3;
  if (n3d > 0) {
    const float* sx = s_xyz + (size_t)k * rows * 3;
    for (int i = tid; i < rows * 3; i += SF_BLOCK) dx[i] = sx[i];
  }
  float4* dk = kp + (size_t)slot * kcap;
  const sf_keypoint* sk = s_kp + (size_t)k * rows;
  for (int i = tid; i < rows; i += SF_BLOCK) {
    const sf_keypoint q = sk[i];
    int o = q.octave & 255;
    o = o < 128 ? o : (-128 | o);
    dk[i] = make_float4(q.x, q.y, __int_as_float(o), 0.f);
  }
  if (tid == 0) meta[slot] = make_int4(rows, n3d > 0 ? rows : 0, rows, cols);
}

// ragged variant for host-buffer batches: per-keyframe table of byte offsets into ONE packed
// staging buffer.  The host packs each keyframe as [descriptors | xyz | keypoints reduced to
// {x, y, raw octave} (12 of the wire's 28 bytes)], every part 16-byte aligned.
struct IngestEntry {
  int32_t rows, cols, n3d, pad;
  uint64_t desc_off, xyz_off, kp_off;   // byte offsets from the start of the staging buffer
};
struct PackedKp { float x, y; int32_t octave; };

__global__ void __launch_bounds__(SF_BLOCK)
k_ingest_ragged(uint32_t* __restrict__ desc, float* __restrict__ xyz, float4* __restrict__ kp, int4* __restrict__ meta,
                int kcap, int w, int first_slot, const IngestEntry* __restrict__ table,
                const uint8_t* __restrict__ stage) {
  const int k = blockIdx.x;
  const int slot = first_slot + k;
  const int tid = threadIdx.x;
  const IngestEntry e = table[k];
  const uint8_t* sd = stage + e.desc_off;
  if ((e.cols & 3) == 0) {
    // dword path (descriptor bytes a multiple of 4; rows start 16-byte aligned in the staging buffer)
    uint32_t* d32 = desc + (size_t)slot * kcap * w;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(sd);
    const int cw = e.cols >> 2;
    for (int i = tid; i < e.rows * w; i += SF_BLOCK) {
      const int r = i / w, b = i - r * w;
      d32[i] = (b < cw) ? s32[(size_t)r * cw + b] : 0u;
    }
  } else {
    uint8_t* d8 = reinterpret_cast<uint8_t*>(desc + (size_t)slot * kcap * w);
    const int rowb = w * 4;
    for (int i = tid; i < e.rows * rowb; i += SF_BLOCK) {
      const int r = i / rowb, b = i - r * rowb;
      d8[i] = (b < e.cols) ? sd[(size_t)r * e.cols + b] : (uint8_t)0;
    }
  }
  if (e.n3d > 0) {
    float* dx = xyz + (size_t)slot * kcap * 3;
    const float* sx = reinterpret_cast<const float*>(stage + e.xyz_off);
    for (int i = tid; i < e.rows * 3; i += SF_BLOCK) dx[i] = sx[i];
  }
  float4* dk = kp + (size_t)slot * kcap;
  const PackedKp* sk = reinterpret_cast<const PackedKp*>(stage + e.kp_off);
  for (int i = tid; i < e.rows; i += SF_BLOCK) {
    const PackedKp q = sk[i];
    int o = q.octave & 255;
    o = o < 128 ? o : (-128 | o);
    dk[i] = make_float4(q.x, q.y, __int_as_float(o), 0.f);
  }
  if (tid == 0) meta[slot] = make_int4(e.rows, e.n3d > 0 ? e.rows : 0, e.rows, e.cols);
}

// Speculative verification (sf_find_matches_and_verify_device): candidate i of the NN filter, (local row r,
// received column c), becomes pair slot i = (slot_other + c, slot_local + r); slots past the candidate count
// (and candidates outside the store) get -1, which every verification kernel answers with a null result.
__global__ void __launch_bounds__(256)
k_spec_pairs(const uint2* __restrict__ cand, const unsigned* __restrict__ count, unsigned grid, int n_l, int n_r,
             int slot_other, int slot_local, int n_slots, int32_t* __restrict__ from, int32_t* __restrict__ to) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= grid) return;
  int f = -1, t = -1;
  if (i < *count) {
    const uint2 rc = cand[i];
    if ((int)rc.x < n_l && (int)rc.y < n_r) {
      f = slot_other + (int)rc.y;
      t = slot_local + (int)rc.x;
      if ((unsigned)f >= (unsigned)n_slots || (unsigned)t >= (unsigned)n_slots) { f = -1; t = -1; }
    }
  }
  from[i] = f;
  to[i] = t;
}

// out[m] = spec[index[m]]: the results of the walk's matches, in walk order (23 x 16 bytes per record)
__global__ void __launch_bounds__(256)
k_spec_gather(const sf_result* __restrict__ spec, const int32_t* __restrict__ index, int n, sf_result* __restrict__ out) {
  static_assert(sizeof(sf_result) % 16 == 0, "sf_result is moved in 16-byte pieces");
  constexpr int PIECES = sizeof(sf_result) / 16;
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int m = g / PIECES, piece = g % PIECES;
  if (m >= n) return;
  const uint4* src = reinterpret_cast<const uint4*>(spec + index[m]);
  reinterpret_cast<uint4*>(out + m)[piece] = src[piece];
}

// Ordered compaction of the accepted results, 1024 candidates per workgroup: k_compact_count leaves the
// number of accepted candidates of every chunk (and the per-candidate flags), k_compact_move lets each
// workgroup sum the counts of the chunks before it (at most a few hundred values) and moves its records
// as 23 x 16 bytes each, consecutive threads taking consecutive pieces.
__global__ void __launch_bounds__(1024)
k_compact_count(const sf_result* __restrict__ res, int n, uint8_t* __restrict__ flags, int32_t* __restrict__ chunk_count) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 1024 + tid;
  const bool ok = i < n && res[i].success != 0;
  if (i < n && flags) flags[i] = ok ? 1 : 0;
  const unsigned long long bal = __ballot(ok);
  if (lane == 0) wsum[wave] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wsum[w];
    chunk_count[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(1024)
k_compact_move(const sf_result* __restrict__ res, int n, sf_result* __restrict__ acc,
               const int32_t* __restrict__ chunk_count, int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int s_dst[1024];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * 1024;
  // offset of this chunk = accepted candidates of all earlier chunks
  int part = 0;
  for (int b = tid; b < (int)blockIdx.x; b += 1024) part += chunk_count[b];
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wsum[w];
    s_base = t;
    if (blockIdx.x == gridDim.x - 1) *total = t + chunk_count[blockIdx.x];
  }
  __syncthreads();
  const int i = base + tid;
  const bool ok = i < n && res[i].success != 0;
  const unsigned long long bal = __ballot(ok);
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(bal);
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) woff += (w < wave) ? wsum[w] : 0;
  s_dst[tid] = ok ? s_base + woff + before : -1;
  __syncthreads();
  const int m = min(1024, n - base);
  for (int e = tid; e < m * 23; e += 1024) {
    const int c = e / 23, piece = e - c * 23;
    const int dst = s_dst[c];
    if (dst >= 0) reinterpret_cast<uint4*>(acc + dst)[piece] = reinterpret_cast<const uint4*>(res + base + c)[piece];
  }
}

}  // namespace

// dwords per stored descriptor row: binary rows 8 (<= 256 bits) or 16; float32 rows (desc_type 1) one per dimension
static int desc_dwords(const sf_context* c, int cols) {
  if (c->params.desc_type == 1) return cols / 4;
  return cols <= 32 ? 8 : 16;
}

static int store_reserve(sf_context* c, Store& s, int slots_needed, int rows, int cols) {
  if (c->params.desc_type == 1) {
    if (cols != 256 && cols != 512)
      return sf_fail(c, SF_ERANGE, "float32 descriptors: %d bytes per row (64 or 128 dimensions = 256 or 512 bytes)", cols);
  } else if (cols < 1 || cols > SF_MAX_DESC_BYTES) {
    return sf_fail(c, SF_ERANGE, "descriptor bytes %d not in 1..%d", cols, SF_MAX_DESC_BYTES);
  }
  if (rows > SF_MAX_FEATURES) return sf_fail(c, SF_ERANGE, "rows %d > int16 limit of KeyPointVec.size", rows);
  const int w = desc_dwords(c, cols);
  int kcap = s.kcap ? s.kcap : std::max(64, (c->params.max_features + 63) & ~63);
  while (kcap < rows) kcap *= 2;
  if (kcap > SF_MAX_KCAP) return sf_fail(c, SF_ERANGE, "%d features per keyframe exceed the kernel capacity %d", rows, SF_MAX_KCAP);
  if (s.slots > 0 && s.w != w) return sf_fail(c, SF_EINVAL, "descriptor width %d B differs from the store's (%d dwords)", cols, s.w);
  int cap = s.cap_slots;
  if (cap < slots_needed) cap = std::max(slots_needed, std::max(cap * 2, &s == &c->store ? c->params.store_capacity : 64));
  if (&s == &c->store) (void)sf_lanes_touch(c, false);     // (a slot of the store is about to be written)
  if (kcap == s.kcap && cap == s.cap_slots && s.w == w) return SF_OK;
  if (&s == &c->store) (void)sf_lanes_touch(c, true);      // the old buffers are freed below: nothing may still read them
  // (re)allocate; keep old contents slot by slot (pitch copy when kcap grew)
  Store n;
  n.kcap = kcap; n.w = w; n.cap_slots = cap; n.slots = s.slots;
  int rc;
  if ((rc = sf_buf_reserve(c, n.desc, (size_t)cap * kcap * w * 4)) != SF_OK ||
      (rc = sf_buf_reserve(c, n.xyz, (size_t)cap * kcap * 12)) != SF_OK ||
      (rc = sf_buf_reserve(c, n.kp, (size_t)cap * kcap * 16)) != SF_OK ||
      (rc = sf_buf_reserve(c, n.meta, (size_t)cap * 16)) != SF_OK) {
    buf_free(n.desc); buf_free(n.xyz); buf_free(n.kp); buf_free(n.meta);   // the old store stays valid
    return rc;
  }
  if (s.slots > 0) {
    SF_HIP(c, hipMemcpy2DAsync(n.desc.p, (size_t)kcap * w * 4, s.desc.p, (size_t)s.kcap * w * 4, (size_t)s.kcap * w * 4, s.slots, hipMemcpyDeviceToDevice, c->stream));
    SF_HIP(c, hipMemcpy2DAsync(n.xyz.p, (size_t)kcap * 12, s.xyz.p, (size_t)s.kcap * 12, (size_t)s.kcap * 12, s.slots, hipMemcpyDeviceToDevice, c->stream));
    SF_HIP(c, hipMemcpy2DAsync(n.kp.p, (size_t)kcap * 16, s.kp.p, (size_t)s.kcap * 16, (size_t)s.kcap * 16, s.slots, hipMemcpyDeviceToDevice, c->stream));
    SF_HIP(c, hipMemcpyAsync(n.meta.p, s.meta.p, (size_t)s.slots * 16, hipMemcpyDeviceToDevice, c->stream));
    SF_HIP(c, hipStreamSynchronize(c->stream));
  }
  buf_free(s.desc); buf_free(s.xyz); buf_free(s.kp); buf_free(s.meta);
  s = n;
  return SF_OK;
}

int sf_launch_ingest(sf_context* c, Store& st, int first_slot, int n, int rows, int cols,
                     const uint8_t* d_desc, const float* d_xyz, const sf_keypoint* d_kp) {
  if (n <= 0) return SF_OK;
  hipLaunchKernelGGL(k_ingest, dim3(n), dim3(SF_BLOCK), 0, c->stream, (uint32_t*)st.desc.p, (float*)st.xyz.p,
                     (float4*)st.kp.p, (int4*)st.meta.p, st.kcap, st.w, first_slot, rows, cols, d_xyz ? rows : 0,
                     d_desc, d_xyz, d_kp);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

static int validate_features(sf_context* c, const sf_features* f) {
  if (!f) return sf_fail(c, SF_EINVAL, "null sf_features");
  if (f->rows > SF_MAX_FEATURES) return sf_fail(c, SF_ERANGE, "rows %d exceed int16", (int)f->rows);
  if (f->rows > 0 && (!f->desc || f->cols == 0)) return sf_fail(c, SF_EINVAL, "descriptors missing");
  if (f->n3d != 0 && f->n3d != (int32_t)f->rows)
    return sf_fail(c, SF_EINVAL, "kpts3D size %d != descriptor rows %d (myRegistrationVis.cpp:859)", f->n3d, (int)f->rows);
  if (f->nkp != (int32_t)f->rows)
    return sf_fail(c, SF_EINVAL, "kpts size %d != descriptor rows %d (myRegistrationVis.cpp:879)", f->nkp, (int)f->rows);
  if (f->n3d > 0 && !f->xyz) return sf_fail(c, SF_EINVAL, "kpts3D missing");
  if (f->nkp > 0 && !f->kpts) return sf_fail(c, SF_EINVAL, "kpts missing");
  return SF_OK;
}

// host features -> one store slot (staged through a device bounce buffer on the handle's stream)
struct Staging {
  Buf &desc, &xyz, &kp;
};
static Staging staging(sf_context* c) { return Staging{c->stage_desc, c->stage_xyz, c->stage_kp}; }

static int store_add_host(sf_context* c, Store& st, const sf_features* f, int* out_slot) {
  int rc = validate_features(c, f);
  if (rc != SF_OK) return rc;
  const int rows = f->rows;
  int cols = f->cols;
  if (rows == 0 && cols == 0) cols = st.slots > 0 ? st.w * 4 : std::max(1, c->params.desc_bytes);
  if ((rc = store_reserve(c, st, st.slots + 1, rows, cols)) != SF_OK) return rc;
  if (st.slots > 0 || rows > 0) {
    // all keyframes of one store share the descriptor width class
    if (st.w != desc_dwords(c, cols)) return sf_fail(c, SF_EINVAL, "descriptor width mismatch");
  }
  Staging sg = staging(c);
  const uint8_t* dd = nullptr; const float* dx = nullptr; const sf_keypoint* dk = nullptr;
  if (rows > 0) {
    if ((rc = sf_buf_reserve(c, sg.desc, (size_t)rows * cols)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, sg.kp, (size_t)rows * sizeof(sf_keypoint))) != SF_OK) return rc;
    SF_HIP(c, hipMemcpyAsync(sg.desc.p, f->desc, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
    SF_HIP(c, hipMemcpyAsync(sg.kp.p, f->kpts, (size_t)rows * sizeof(sf_keypoint), hipMemcpyHostToDevice, c->stream));
    dd = (const uint8_t*)sg.desc.p; dk = (const sf_keypoint*)sg.kp.p;
    if (f->n3d > 0) {
      if ((rc = sf_buf_reserve(c, sg.xyz, (size_t)rows * 12)) != SF_OK) return rc;
      SF_HIP(c, hipMemcpyAsync(sg.xyz.p, f->xyz, (size_t)rows * 12, hipMemcpyHostToDevice, c->stream));
      dx = (const float*)sg.xyz.p;
    }
  }
  if ((rc = sf_launch_ingest(c, st, st.slots, 1, rows, cols, dd, dx, dk)) != SF_OK) return rc;
  // the bounce buffers are reused by the next call: drain before returning
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (out_slot) *out_slot = st.slots;
  st.slots += 1;
  return SF_OK;
}

// n host keyframes -> consecutive slots.  Features are packed into ONE pinned staging buffer in
// chunks of a few MB: worker threads pack chunk k+1 while the H2D copy of chunk k is in flight, then
// ONE ragged ingest launch converts everything (no per-keyframe synchronisation).  Small batches
// (a single service call) are packed inline by the calling thread.
static inline size_t pad16(size_t v) { return (v + 15) & ~(size_t)15; }

static void pack_keyframe(uint8_t* hp, const IngestEntry& e, const sf_features* f) {
  if (f->rows == 0) return;
  memcpy(hp + e.desc_off, f->desc, (size_t)f->rows * f->cols);
  if (f->n3d > 0) memcpy(hp + e.xyz_off, f->xyz, (size_t)f->rows * 12);
  PackedKp* k = reinterpret_cast<PackedKp*>(hp + e.kp_off);
  const sf_keypoint* src = f->kpts;
  for (int i = 0; i < (int)f->rows; ++i) { k[i].x = src[i].x; k[i].y = src[i].y; k[i].octave = src[i].octave; }
}


// Packing workers of the host-buffer batch ingest: started once per handle (first large batch), parked on a
// condition variable between batches, joined by sf_destroy.  A batch is cut into chunks; worker t packs the
// keyframes lo + t, lo + t + workers, ... of every chunk in turn and bumps done[k]; the calling thread ships
// chunk k to the device as soon as every worker has passed it.
struct IngestPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_start, cv_idle;
  uint64_t generation = 0;
  int running = 0;
  bool quit = false;
  // the job of the current generation
  uint8_t* hp = nullptr;
  const IngestEntry* tab = nullptr;
  const sf_features* const* feats = nullptr;
  const std::pair<size_t, int>* chunks = nullptr;
  int n_chunks = 0;
  std::vector<std::atomic<int>> done;
  std::vector<IngestEntry> tab_storage;
  std::vector<std::pair<size_t, int>> chunk_storage;

  explicit IngestPool(int workers) : done(0) {
    for (int t = 0; t < workers; ++t) threads.emplace_back([this, t]() { work(t); });
  }
  ~IngestPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_start.notify_all();
    for (auto& th : threads) th.join();
  }
  void work(int t) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_start.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
      }
      const int nw = (int)threads.size();
      int lo = 0;
      for (int k = 0; k < n_chunks; ++k) {
        const int hi = chunks[k].second;
        for (int i = lo + t; i < hi; i += nw) pack_keyframe(hp, tab[i], feats[i]);
        done[k].fetch_add(1, std::memory_order_release);
        lo = hi;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) cv_idle.notify_all();
      }
    }
  }
  void start() {
    if ((int)done.size() < n_chunks) done = std::vector<std::atomic<int>>(n_chunks);
    for (int k = 0; k < n_chunks; ++k) done[k].store(0, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(mu);
      running = (int)threads.size();
      ++generation;
    }
    cv_start.notify_all();
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu);
    cv_idle.wait(lk, [&] { return running == 0; });
  }
};

void sf_ingest_pool_destroy(sf_context* c) {
  delete c->ingest_pool;
  c->ingest_pool = nullptr;
}

static int store_add_host_batch(sf_context* c, Store& st, const sf_features* const* feats, int n, int* first_slot) {
  if (n <= 0) return SF_OK;
  int rc;
  int max_rows = 0, cols = 0;
  for (int i = 0; i < n; ++i) {
    if ((rc = validate_features(c, feats[i])) != SF_OK) return rc;
    max_rows = std::max<int>(max_rows, feats[i]->rows);
    if (feats[i]->rows > 0) {
      if (cols == 0) cols = feats[i]->cols;
      if (desc_dwords(c, feats[i]->cols) != desc_dwords(c, cols)) return sf_fail(c, SF_EINVAL, "descriptor width classes differ inside one batch");
      if (feats[i]->cols > (c->params.desc_type == 1 ? SF_MAX_DESC_BYTES_F32 : SF_MAX_DESC_BYTES))
        return sf_fail(c, SF_ERANGE, "descriptor bytes %d > %d", (int)feats[i]->cols, c->params.desc_type == 1 ? SF_MAX_DESC_BYTES_F32 : SF_MAX_DESC_BYTES);
    }
  }
  if (cols == 0) cols = st.slots > 0 ? st.w * 4 : std::max(1, c->params.desc_bytes);
  if ((rc = store_reserve(c, st, st.slots + n, max_rows, cols)) != SF_OK) return rc;

  // layout: [table][keyframe 0: desc | xyz | kp][keyframe 1 ...], chunk boundaries every ~4 MB
  // (table / chunk scratch lives in the pool object when there is one, else on this call's stack vectors)
  std::vector<IngestEntry> tab_local;
  std::vector<std::pair<size_t, int>> chunks_local;   // (end offset, end keyframe)
  std::vector<IngestEntry>& tab = c->ingest_pool ? c->ingest_pool->tab_storage : tab_local;
  std::vector<std::pair<size_t, int>>& chunks = c->ingest_pool ? c->ingest_pool->chunk_storage : chunks_local;
  tab.resize(n);
  chunks.clear();
  const size_t tb = pad16((size_t)n * sizeof(IngestEntry));
  const size_t chunk_bytes = (size_t)4 << 20;
  size_t off = tb, chunk_start = tb;
  for (int i = 0; i < n; ++i) {
    const sf_features* f = feats[i];
    IngestEntry& e = tab[i];
    e.rows = f->rows; e.cols = f->rows > 0 ? f->cols : cols; e.n3d = f->n3d; e.pad = 0;
    e.desc_off = off; off += pad16((size_t)f->rows * f->cols);
    e.xyz_off = off;  off += f->n3d > 0 ? pad16((size_t)f->rows * 12) : 0;
    e.kp_off = off;   off += pad16((size_t)f->rows * sizeof(PackedKp));
    if (off - chunk_start >= chunk_bytes || i == n - 1) { chunks.push_back({off, i + 1}); chunk_start = off; }
  }
  const size_t total = off;
  // pinned host staging owned by the handle (released by sf_destroy)
  if (total > c->ingest_pinned_bytes) {
    if (c->ingest_pinned) (void)hipHostFree(c->ingest_pinned);
    c->ingest_pinned = nullptr;
    c->ingest_pinned_bytes = 0;
    const size_t want = total + total / 2;
    hipError_t e = hipHostMalloc(&c->ingest_pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) -> %s", want, hipGetErrorString(e));
    c->ingest_pinned_bytes = want;
  }
  if ((rc = sf_buf_reserve(c, c->stage_desc, total)) != SF_OK) return rc;
  uint8_t* hp = (uint8_t*)c->ingest_pinned;
  uint8_t* dp = (uint8_t*)c->stage_desc.p;
  memcpy(hp, tab.data(), (size_t)n * sizeof(IngestEntry));

  const int n_chunks = (int)chunks.size();
  int workers = 0;
  if (n_chunks >= 2) workers = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
  hipError_t herr = hipSuccess;
  if (workers <= 1) {
    for (int i = 0; i < n; ++i) pack_keyframe(hp, tab[i], feats[i]);
    herr = hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, c->stream);
  } else {
    if (!c->ingest_pool) {
      // first large batch of this handle: start the workers, and move the table / chunk lists into the pool
      c->ingest_pool = new IngestPool(workers);
      c->ingest_pool->tab_storage.swap(tab_local);
      c->ingest_pool->chunk_storage.swap(chunks_local);
    }
    IngestPool& pool = *c->ingest_pool;
    pool.hp = hp;
    pool.tab = pool.tab_storage.data();
    pool.feats = feats;
    pool.chunks = pool.chunk_storage.data();
    pool.n_chunks = n_chunks;
    pool.start();
    const int nw = (int)pool.threads.size();
    size_t sent = 0;
    for (int k = 0; k < n_chunks; ++k) {
      while (pool.done[k].load(std::memory_order_acquire) < nw) std::this_thread::yield();
      const size_t end = pool.chunk_storage[k].first;
      if (herr == hipSuccess) herr = hipMemcpyAsync(dp + sent, hp + sent, end - sent, hipMemcpyHostToDevice, c->stream);
      sent = end;
    }
    pool.wait_idle();
  }
  if (herr != hipSuccess) return sf_fail(c, SF_EHIP, "staging H2D copy -> %s", hipGetErrorString(herr));
  hipLaunchKernelGGL(k_ingest_ragged, dim3(n), dim3(SF_BLOCK), 0, c->stream, (uint32_t*)st.desc.p, (float*)st.xyz.p,
                     (float4*)st.kp.p, (int4*)st.meta.p, st.kcap, st.w, st.slots, (const IngestEntry*)dp, dp);
  SF_HIP(c, hipGetLastError());
  // the pinned staging is reused by the next call: the copies must have left host memory
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (first_slot) *first_slot = st.slots;
  st.slots += n;
  return SF_OK;
}

// ---- parameters ---------------------------------------------------------------------------------
extern "C" int sf_abi_version(void) { return SF_ABI_VERSION; }

extern "C" void sf_default_params(sf_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->netvlad_distance = 0.13;        // multi_robot_separators.launch:19
  p->netvlad_dimensions = 128;       // :20
  p->netvlad_max_matches_nb = 20;    // :22
  p->nn_precision = 1;               // fp16 filter + exact f64 refinement (identical matches)
  p->min_inliers = 5;                // :23 separators_min_inliers
  p->inlier_distance = 0.1f;         // rtabmap Vis/InlierDistance [upstream default]
  p->iterations = 300;               // Vis/Iterations
  p->refine_iterations = 5;          // Vis/RefineIterations
  p->refine_sigma = 3.0;
  p->estimation_type = 0;            // 3D->3D (BASELINE.json north_star)
  p->nndr = 0.6f;                    // Vis/CorNNDR
  p->guess_win_size = 20;            // Vis/CorGuessWinSize
  p->ransac_adaptive_stop = 1;
  p->max_sample_checks = 1000;
  p->seed = 12345;
  const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  memcpy(p->local_transform, I, sizeof(I));
  p->store_capacity = 1024;
  p->max_features = 512;
  p->desc_bytes = 32;
  p->pnp_reproj_error = 2.0f;        // Vis/PnPReprojError
  p->pnp_flags = 0;                  // Vis/PnPFlags (cv::SOLVEPNP_ITERATIVE)
  p->pnp_refine_iterations = 0;      // Vis/PnPRefineIterations
  p->bundle_adjustment = 0;          // Vis/BundleAdjustment (rtabmap: 1 with g2o; off here: north_star's path has none)
  p->ba_iterations = 20;             // Optimizer/Iterations
  p->ba_robust_kernel_delta = 8.f;   // g2o/RobustKernelDelta
  p->ba_pixel_variance = 1.f;        // g2o/PixelVariance
  p->stereo_baseline = 0.f;
  p->force_3dof = 0;                 // Reg/Force3DoF
  p->forward_est_only = 1;           // Vis/ForwardEstOnly
  p->desc_type = 0;                  // binary descriptors (the reference's wire carries nothing else)
}

static int fill_device_params(sf_context* c) {
  const sf_params& p = c->params;
  if (p.estimation_type != 0 && p.estimation_type != 1)
    return sf_fail(c, SF_EINVAL, "estimation_type %d not implemented (0 = 3D->3D, 1 = PnP)", p.estimation_type);
  if (p.estimation_type == 1) {
    if (p.pnp_flags != 0) return sf_fail(c, SF_EINVAL, "pnp_flags %d not implemented (0 = SOLVEPNP_ITERATIVE only)", p.pnp_flags);
    if (p.pnp_refine_iterations < 0) return sf_fail(c, SF_EINVAL, "pnp_refine_iterations must be >= 0");
    if (!(p.pnp_reproj_error > 0.f)) return sf_fail(c, SF_EINVAL, "pnp_reproj_error must be > 0");
  }
  if (p.min_inliers < 1) return sf_fail(c, SF_EINVAL, "min_inliers must be >= 1 (myRegistrationVis.cpp:117)");
  if (!(p.inlier_distance > 0.f)) return sf_fail(c, SF_EINVAL, "inlier_distance must be > 0 (:118)");
  if (p.iterations < 1) return sf_fail(c, SF_EINVAL, "iterations must be > 0 (:119)");
  if (p.iterations > 30000) return sf_fail(c, SF_ERANGE, "iterations > 30000");
  if (p.max_sample_checks < 1) return sf_fail(c, SF_EINVAL, "max_sample_checks must be >= 1");
  if (p.netvlad_max_matches_nb < 0) return sf_fail(c, SF_EINVAL, "netvlad_max_matches_nb < 0");
  if (p.bundle_adjustment != 0) {
    if (p.bundle_adjustment != 1) return sf_fail(c, SF_EINVAL, "bundle_adjustment %d not implemented (0 = off, 1 = on)", p.bundle_adjustment);
    if (!(p.image_width > 0 && p.image_height > 0 && p.fx > 0.0 && p.fy > 0.0))
      return sf_fail(c, SF_EINVAL, "bundle adjustment needs a calibrated camera (myRegistrationVis.cpp:1230 UASSERT)");
    if (p.ba_iterations < 0 || !(p.ba_pixel_variance > 0.f) || !(p.ba_robust_kernel_delta > 0.f) || !(p.stereo_baseline >= 0.f))
      return sf_fail(c, SF_EINVAL, "bundle adjustment: ba_iterations >= 0, ba_pixel_variance > 0, ba_robust_kernel_delta > 0, stereo_baseline >= 0");
  }
  if (p.desc_type != 0 && p.desc_type != 1) return sf_fail(c, SF_EINVAL, "desc_type %d unknown (0 = binary rows, 1 = float32 rows)", p.desc_type);
  if (p.desc_type == 1 && p.desc_bytes != 256 && p.desc_bytes != 512 && p.desc_bytes != 32)
    return sf_fail(c, SF_EINVAL, "desc_type 1: desc_bytes %d (float32 rows of 64 or 128 dimensions: 256 or 512)", p.desc_bytes);
  // float32 descriptors with the width left at sf_default_params' 32 (a binary width): 64 dimensions.  The width only
  // matters before the first non-empty keyframe arrives -- an EMPTY first keyframe (the reference tolerates them: the
  // fake-words path) takes it for its row pitch, and 32 bytes per float row was refused by the store (round 4).
  if (p.desc_type == 1 && p.desc_bytes == 32) c->params.desc_bytes = 256;
  DeviceParams& d = c->dparams;
  memset(&d, 0, sizeof(d));
  d.force_3dof = p.force_3dof != 0;
  d.bidirectional = p.forward_est_only == 0;
  d.nndr = p.nndr;
  d.min_inliers = p.min_inliers;
  d.iterations = p.iterations;
  d.refine_iterations = p.refine_iterations;
  d.refine_sigma = p.refine_sigma;
  d.inlier_thr = (double)p.inlier_distance;
  d.adaptive_stop = p.ransac_adaptive_stop;
  d.max_sample_checks = p.max_sample_checks;
  d.seed = p.seed;
  d.guess_win = p.guess_win_size;
  d.calibrated = (p.image_width > 0 && p.image_height > 0 && p.fx > 0.0 && p.fy > 0.0) ? 1 : 0;
  d.fx = p.fx; d.fy = p.fy; d.cx = p.cx; d.cy = p.cy;
  d.wlim = (float)(p.image_width - 1);
  d.hlim = (float)(p.image_height - 1);
  memcpy(d.L, p.local_transform, sizeof(d.L));
  d.estimation_type = p.estimation_type;
  d.pnp_reproj_error = p.pnp_reproj_error;
  d.pnp_refine_iterations = p.pnp_refine_iterations;
  d.bundle_adjustment = p.bundle_adjustment;
  d.ba_iterations = p.ba_iterations;
  d.ba_robust_kernel_delta = p.ba_robust_kernel_delta;
  d.ba_pixel_variance = p.ba_pixel_variance;
  d.stereo_baseline = p.stereo_baseline;
  {
    const double thr = (double)p.pnp_reproj_error;
    d.pnp_thr2f = (float)(thr * thr);     // OpenCV: float t = (float)(thresh*thresh); err <= t
  }
  if (const char* v = getenv("SF_RANSAC_STOP")) d.dbg_stop = atoi(v);
  {
    // grid for the guided pass: cell >= window radius, at most 48 x 48 cells
    const float w = (float)std::max(p.image_width, 1), h = (float)std::max(p.image_height, 1);
    float cell = std::max((float)std::max(p.guess_win_size, 1), std::max(ceilf(w / 48.f), ceilf(h / 48.f)));
    cell *= 1.001f;   // strictly larger than the padded reach used by the kernel
    d.grid_gx = std::max(1, std::min(48, (int)ceilf(w / cell)));
    d.grid_gy = std::max(1, std::min(48, (int)ceilf(h / cell)));
    d.grid_inv_cell = 1.f / cell;
  }
  return SF_OK;
}

// ---- lifecycle ----------------------------------------------------------------------------------
extern "C" int sf_create(const sf_params* p, int device, sf_handle* out) {
  if (!out) return SF_EINVAL;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return sf_fail(nullptr, SF_ENODEV, "no HIP device visible (%s); this library has no CPU fallback",
                   e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return sf_fail(nullptr, SF_EINVAL, "device %d out of range (%d visible)", device, ndev);
  sf_context* c = new (std::nothrow) sf_context();
  if (!c) return SF_ENOMEM;
  if (p) c->params = *p; else sf_default_params(&c->params);
  c->device = device;
  int rc = fill_device_params(c);
  if (rc != SF_OK) { g_create_error = c->err; delete c; return rc; }
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    sf_fail(nullptr, SF_EHIP, "device init -> %s", hipGetErrorString(e));
    delete c;
    return SF_EHIP;
  }
  c->own_stream = true;
  if (const char* v = getenv("SF_MATCH_VARIANT")) c->match_variant = atoi(v);
  if (const char* v = getenv("SF_FUSED")) {   // 0: stage kernels (A/B reference), 1: fused kernel, 2: split pipeline
    c->fused = atoi(v) != 0;
    c->split = atoi(v) == 2;
  }
  if (const char* v = getenv("SF_STEP_SPLIT_MIN")) c->split_auto_min = std::max(1, atoi(v));
  if (const char* v = getenv("SF_STEP_SPLIT")) c->split_auto = atoi(v) != 0;   // 0: overlapped steps keep the fused kernel
  if (const char* v = getenv("SF_CHAIN_PNP")) c->chain_pnp = atoi(v) != 0;      // 0: PnP on the five stage launches
  if (const char* v = getenv("SF_BA_OCC")) c->ba_occ = atoi(v) == 1 ? 1 : atoi(v) == 2 ? 2 : 0;
  if (const char* v = getenv("SF_BA_NW")) { const int nw = atoi(v); c->ba_nw = (nw == 1 || nw == 2) ? nw : 4; }
  if (const char* v = getenv("SF_CHAIN_PNP_NW")) { const int nw = atoi(v); c->chain_pnp_nw = (nw == 1 || nw == 2) ? nw : 4; }
  if (const char* v = getenv("SF_CHAIN_NW")) { const int nw = atoi(v); c->chain_nw = (nw == 1 || nw == 2) ? nw : 4; }
  if (const char* v = getenv("SF_MATCH_MFMA")) c->match_mfma = atoi(v) != 0;   // 0: VALU matcher (A/B reference)
  if (const char* v = getenv("SF_DEBUG_CORR")) c->debug_corr = atoi(v) != 0;   // 1: correspondence lists kept in HBM
  if (const char* v = getenv("SF_OVERLAP")) c->overlap = atoi(v) != 0;         // 1: two-stream halves (verify_device)
  if (const char* v = getenv("SF_STEP_OVERLAP")) c->step_overlap = atoi(v) != 0;   // 1: SF_OPT_STEP_OVERLAP from the start
  if (const char* v = getenv("SF_OVERLAP_MIN")) c->overlap_min_pairs = std::max(2, atoi(v));
  if (const char* v = getenv("SF_STEP_DEPTH")) c->step_depth = std::max(1, std::min(SF_STEP_MAX_DEPTH, atoi(v)));
  if (const char* v = getenv("SF_STEP_LANES")) c->step_lanes = std::max(1, std::min(SF_STEP_MAX_LANES, atoi(v)));
  if (const char* v = getenv("SF_STEP_SPECULATE")) c->step_speculate = atoi(v) != 0;   // 0: every device step in the serial form
  if (const char* v = getenv("SF_STEP_DEVICE_WALK")) c->step_device_walk = atoi(v) != 0;   // 0: round 3's host walk inside sf_step_issue
  if ((rc = sf_buf_reserve(c, c->counters, 64)) != SF_OK) { g_create_error = c->err; sf_destroy(c); return rc; }
  *out = c;
  return SF_OK;
}

extern "C" void sf_destroy(sf_handle c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->twin) {   // shadow context of the two-stream verification: workspace, counters, stream only
    sf_context* t = c->twin;
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    prof_resolve(t);
    for (hipEvent_t e : t->prof_event_pool) (void)hipEventDestroy(e);
    Buf* tb[] = {&t->corr1, &t->corr2, &t->hdr1, &t->hdr2, &t->pass1, &t->pass2, &t->list1, &t->list3, &t->counters, &t->flags};
    for (Buf* b : tb) buf_free(*b);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    c->twin = nullptr;
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  (void)sf_comm_destroy(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  prof_resolve(c);
  for (hipEvent_t e : c->prof_event_pool) (void)hipEventDestroy(e);
  Buf* bufs[] = {&c->store.desc, &c->store.xyz, &c->store.kp, &c->store.meta, &c->scratch.desc, &c->scratch.xyz,
                 &c->scratch.kp, &c->scratch.meta, &c->pair_from, &c->pair_to, &c->corr1, &c->corr2, &c->hdr1,
                 &c->hdr2, &c->pass1, &c->pass2, &c->list1, &c->list3, &c->counters, &c->results,
                 &c->flags, &c->nn_local.rows, &c->nn_local.norms, &c->nn_local.rows_h, &c->nn_local.norms_k, &c->nn_recv.norms_k, &c->nn_recv.rows,
                 &c->nn_recv.norms, &c->nn_recv.rows_h, &c->d_mask_local, &c->d_mask_other, &c->d_ign_ptr,
                 &c->d_ign_col, &c->nn_rowmin, &c->nn_exact, &c->nn_cand, &c->nn_scalar, &c->comm_scratch, &c->compact_scratch, &c->trace, &c->stage_desc, &c->stage_xyz, &c->stage_kp,
                 &c->ex_integral, &c->ex_desc, &c->ex_xyz, &c->ex_keep, &c->ex_rows, &c->brief_tests,
                 &c->gf_planes, &c->gf_keys, &c->gf_tmp, &c->gf_lists, &c->gf_scalar, &c->lk_pyr, &c->ft_images, &c->ft_kpts, &c->ft_flow, &c->ft_wire,
                 &c->ft_counts, &c->pass_back, &c->dir_mask};
  for (Buf* b : bufs) buf_free(*b);
  for (sf_context::StepLane& L : c->lanes) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    Buf* lb[] = {&L.pair_from, &L.pair_to, &L.corr1, &L.corr2, &L.hdr1, &L.hdr2, &L.pass1, &L.pass2, &L.pass_back, &L.dir_mask,
                 &L.list1, &L.list3, &L.counters, &L.results, &L.flags, &L.nn_cand, &L.spec_from, &L.spec_to, &L.spec_results,
                 &L.spec_index, &L.compact_scratch, &L.step_nn, &L.walk_scratch};
    for (Buf* b : lb) buf_free(*b);
    if (L.ev_main) (void)hipEventDestroy(L.ev_main);
    if (L.ev_filter) (void)hipEventDestroy(L.ev_filter);
    if (L.ev_walk) (void)hipEventDestroy(L.ev_walk);
    if (L.aux) { (void)hipStreamSynchronize(L.aux); (void)hipStreamDestroy(L.aux); }
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  if (c->ev_filter) (void)hipEventDestroy(c->ev_filter);
  if (c->ev_walk) (void)hipEventDestroy(c->ev_walk);
  if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
  for (int k = 0; k < SF_STEP_MAX_LANES; ++k) {                 // (placed streams nobody asked for)
    if (c->placement.main[k]) (void)hipStreamDestroy(c->placement.main[k]);
    if (c->placement.aux[k]) (void)hipStreamDestroy(c->placement.aux[k]);
  }
  if (c->placement.copy) (void)hipStreamDestroy(c->placement.copy);
  buf_free(c->step_nn);
  buf_free(c->walk_scratch);
  if (c->ev_prep) (void)hipEventDestroy(c->ev_prep);
  sf_netvlad_free(c);
  sf_ingest_pool_destroy(c);
  if (c->ingest_pinned) (void)hipHostFree(c->ingest_pinned);
  if (c->nn_stage_pinned) (void)hipHostFree(c->nn_stage_pinned);
  if (c->nn_stage_done) (void)hipEventDestroy(c->nn_stage_done);
  buf_free(c->nn_stage_dev);
  if (c->nn_pinned) (void)hipHostFree(c->nn_pinned);
  if (c->pairs_pinned) (void)hipHostFree(c->pairs_pinned);
  if (c->count_pinned) (void)hipHostFree(c->count_pinned);
  if (c->pairs_staged) (void)hipEventDestroy(c->pairs_staged);
  {
    Buf* sb[] = {&c->spec_from, &c->spec_to, &c->spec_results, &c->spec_index};
    for (Buf* b : sb) buf_free(*b);
    if (c->spec_index_pinned) (void)hipHostFree(c->spec_index_pinned);
    if (c->spec_index_staged) (void)hipEventDestroy(c->spec_index_staged);
    if (c->spec.ev_refined) (void)hipEventDestroy(c->spec.ev_refined);
    if (c->spec.ev_copied) (void)hipEventDestroy(c->spec.ev_copied);
    if (c->spec.copy_stream) { (void)hipStreamSynchronize(c->spec.copy_stream); (void)hipStreamDestroy(c->spec.copy_stream); }
  }
  for (auto& sb : c->step_blocks) {
    if (sb.pinned) (void)hipHostFree(sb.pinned);
    if (sb.done) (void)hipEventDestroy(sb.done);
    if (sb.copied) (void)hipEventDestroy(sb.copied);
    buf_free(sb.dev);
    buf_free(sb.dev_records);
  }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

#ifdef SF_CHAIN_TRACE
// diagnostic build only: phase timestamps of the last verification ([n][32] uint64, 100 MHz ticks)
extern "C" int sf_debug_chain_trace(sf_handle c, unsigned long long* out, int32_t n) {
  if (!c || !out || n < 0 || n > c->ws_pairs) return SF_EINVAL;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  SF_HIP(c, hipMemcpy(out, c->trace.p, (size_t)n * SF_TRACE_SLOTS * 8, hipMemcpyDeviceToHost));
  return SF_OK;
}
#endif

extern "C" const char* sf_last_error(sf_handle c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" int sf_get_params(sf_handle c, sf_params* out) {
  if (!c || !out) return SF_EINVAL;
  *out = c->params;
  return SF_OK;
}

extern "C" int sf_set_stream(sf_handle c, void* hip_stream) {
  if (!c) return SF_EINVAL;
  (void)sf_lanes_touch(c, true);
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  return SF_OK;
}

extern "C" int sf_synchronize(sf_handle c) {
  if (!c) return SF_EINVAL;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  for (auto& L : c->lanes) {
    if (L.stream) SF_HIP(c, hipStreamSynchronize(L.stream));
    if (L.aux) SF_HIP(c, hipStreamSynchronize(L.aux));
  }
  if (c->aux) SF_HIP(c, hipStreamSynchronize(c->aux));
  return SF_OK;
}

// ---- keyframe store -----------------------------------------------------------------------------
extern "C" int sf_store_add_keyframe(sf_handle c, const sf_features* f, int32_t* out_slot) {
  if (!c) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return store_add_host(c, c->store, f, out_slot);
}

extern "C" int sf_store_add_keyframes_device(sf_handle c, int32_t n, int32_t rows, int32_t cols,
                                             const uint8_t* d_desc, const float* d_xyz,
                                             const sf_keypoint* d_kp, int32_t* out_first_slot) {
  if (!c || n < 0 || rows < 0) return SF_EINVAL;
  if (n == 0) return SF_OK;
  if (rows > 0 && (!d_desc || !d_kp)) return sf_fail(c, SF_EINVAL, "device descriptor / keypoint pointers missing");
  SF_HIP(c, hipSetDevice(c->device));
  int rc = store_reserve(c, c->store, c->store.slots + n, rows, cols);
  if (rc != SF_OK) return rc;
  if ((rc = sf_launch_ingest(c, c->store, c->store.slots, n, rows, cols, d_desc, d_xyz, d_kp)) != SF_OK) return rc;
  if (out_first_slot) *out_first_slot = c->store.slots;
  c->store.slots += n;
  return SF_OK;
}

// ---- feature extraction (kernels in k_extract.hip) ----------------------------------
static int brief_upload(sf_context* c) {
  int rc;
  if ((rc = sf_buf_reserve(c, c->brief_tests, sizeof c->brief_host)) != SF_OK) return rc;
  SF_HIP(c, hipMemcpyAsync(c->brief_tests.p, c->brief_host, (size_t)c->brief_bytes * 32, hipMemcpyHostToDevice, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));   // (the host table may change right after the call returns)
  return SF_OK;
}

extern "C" int sf_brief_set_pattern(sf_handle c, const int8_t* tests, int32_t bytes) {
  if (!c || !tests) return SF_EINVAL;
  if (bytes != 16 && bytes != 32 && bytes != 64) return sf_fail(c, SF_ERANGE, "BRIEF descriptors are 16, 32 or 64 bytes, not %d", bytes);
  for (int t = 0; t < bytes * 32; ++t)
    if (tests[t] < -24 || tests[t] > 24) return sf_fail(c, SF_ERANGE, "BRIEF test offset %d outside the 48 px patch", (int)tests[t]);
  SF_HIP(c, hipSetDevice(c->device));
  memcpy(c->brief_host, tests, (size_t)bytes * 32);
  c->brief_bytes = bytes;
  return brief_upload(c);
}

static int brief_ensure(sf_context* c) {
  if (c->params.desc_type != 0)
    return sf_fail(c, SF_EINVAL, "the feature extraction writes BRIEF (binary) descriptors: a handle with desc_type %d cannot store them", c->params.desc_type);
  if (c->brief_bytes) return SF_OK;
  const int want = c->params.desc_bytes;
  c->brief_bytes = (want == 16 || want == 64) ? want : 32;
  sf_brief_default_pattern(c->brief_host, c->brief_bytes);
  return brief_upload(c);
}

extern "C" int sf_brief_get_pattern(sf_handle c, int8_t* tests, int32_t cap_bytes, int32_t* bytes) {
  if (!c || !bytes) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  int rc = brief_ensure(c);
  if (rc != SF_OK) return rc;
  *bytes = c->brief_bytes;
  if (tests) {
    if (cap_bytes < c->brief_bytes) return sf_fail(c, SF_ERANGE, "pattern buffer holds %d of %d descriptor bytes", cap_bytes, c->brief_bytes);
    memcpy(tests, c->brief_host, (size_t)c->brief_bytes * 32);
  }
  return SF_OK;
}

extern "C" int sf_netvlad_load(sf_handle c, const sf_netvlad_weights* w) {
  if (!c) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return sf_netvlad_load_impl(c, w);
}

extern "C" int sf_netvlad_infer_device(sf_handle c, const float* d_image_rgb, int32_t width, int32_t height, float* d_out,
                                       int32_t n_out) {
  if (!c || !d_image_rgb || !d_out) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return sf_netvlad_infer_impl(c, d_image_rgb, height, width, d_out, n_out);
}

extern "C" int sf_netvlad_infer_batch_device(sf_handle c, const float* d_images_rgb, int32_t n_images, int32_t width,
                                             int32_t height, float* d_out, int32_t n_out) {
  if (!c || !d_images_rgb || !d_out) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return sf_netvlad_infer_batch_impl(c, d_images_rgb, n_images, height, width, d_out, n_out);
}

extern "C" int sf_detect_corners_device(sf_handle c, const uint8_t* d_image, int32_t width, int32_t height, int32_t pitch,
                                        int32_t max_corners, double quality_level, double min_distance,
                                        sf_keypoint* d_kpts_out, int32_t cap, int32_t* n_out) {
  if (!c || !n_out || cap < 0 || (cap > 0 && !d_kpts_out)) return SF_EINVAL;
  *n_out = 0;
  if (!d_image || width < 3 || height < 3 || pitch < width)
    return sf_fail(c, SF_EINVAL, "image missing or malformed (%d x %d, pitch %d)", width, height, pitch);
  if (!(quality_level > 0.0) || !(min_distance >= 0.0))
    return sf_fail(c, SF_EINVAL, "qualityLevel must be > 0 and minDistance >= 0 (cv::goodFeaturesToTrack asserts the same)");
  if ((long long)width * height > (1ll << 26)) return sf_fail(c, SF_ERANGE, "image of %d x %d pixels is too large", width, height);
  SF_HIP(c, hipSetDevice(c->device));
  return sf_launch_detect_corners(c, d_image, width, height, pitch, max_corners, quality_level, min_distance, d_kpts_out,
                                  cap, n_out);
}

extern "C" void sf_stereo_flow_defaults(sf_stereo_flow_params* p) {
  if (!p) return;
  p->win_width = 15; p->win_height = 3;        // Stereo/WinWidth, Stereo/WinHeight [upstream rtabmap Parameters.h]
  p->max_level = 5;                            // Stereo/MaxLevel
  p->iterations = 30;                          // Stereo/Iterations
  p->epsilon = 0.01;                           // Stereo/Eps
  p->min_disparity = 0.5f; p->max_disparity = 128.0f;
  p->min_eig_threshold = 1e-4f;                // the literal in StereoOpticalFlow::computeCorrespondences
}

extern "C" int sf_stereo_correspondences_device(sf_handle c, const uint8_t* d_left, const uint8_t* d_right, int32_t width,
                                                int32_t height, int32_t pitch, const sf_keypoint* d_kpts, int32_t n,
                                                const sf_stereo_flow_params* params, float* d_right_xy,
                                                uint8_t* d_status, float* d_right_x, float* d_err) {
  if (!c || n < 0) return SF_EINVAL;
  if (!d_left || !d_right || width < 1 || height < 1 || pitch < width)
    return sf_fail(c, SF_EINVAL, "stereo pair missing or malformed (%d x %d, pitch %d)", width, height, pitch);
  if (n > 0 && (!d_kpts || !d_right_xy || !d_status)) return sf_fail(c, SF_EINVAL, "corners or output arrays missing");
  sf_stereo_flow_params prm;
  if (params) prm = *params; else sf_stereo_flow_defaults(&prm);
  if (prm.win_width <= 2 || prm.win_height <= 2)
    return sf_fail(c, SF_EINVAL, "window of %d x %d: both sides must be > 2 (cv::calcOpticalFlowPyrLK asserts the same)", prm.win_width, prm.win_height);
  if ((long long)prm.win_width * prm.win_height > 1024) return sf_fail(c, SF_ERANGE, "window of %d x %d exceeds 1024 pixels", prm.win_width, prm.win_height);
  if (prm.max_level < 0 || prm.max_level > 15) return sf_fail(c, SF_ERANGE, "max_level %d outside 0 .. 15", prm.max_level);
  if (!(prm.epsilon == prm.epsilon)) return sf_fail(c, SF_EINVAL, "epsilon is NaN");
  if (n == 0) return SF_OK;
  SF_HIP(c, hipSetDevice(c->device));
  return sf_launch_stereo_flow(c, d_left, d_right, width, height, pitch, d_kpts, n, &prm, d_right_xy, d_status, d_right_x, d_err);
}

extern "C" int sf_extract_keyframe_device(sf_handle c, const uint8_t* d_left, int32_t width, int32_t height,
                                          int32_t pitch, const sf_keypoint* d_kpts, const float* d_right_x,
                                          const uint8_t* d_status, int32_t n, const sf_stereo_camera* cam,
                                          int32_t* out_slot, int32_t* out_rows, uint8_t* d_desc_out,
                                          float* d_xyz_out, sf_keypoint* d_kpts_out) {
  if (!c || !cam || n < 0) return SF_EINVAL;
  if (!d_left || width < 1 || height < 1 || pitch < width) return sf_fail(c, SF_EINVAL, "left image missing or malformed (%d x %d, pitch %d)", width, height, pitch);
  if (n > 0 && !d_kpts) return sf_fail(c, SF_EINVAL, "keypoints missing");
  if (n > SF_MAX_FEATURES) return sf_fail(c, SF_ERANGE, "%d corners > int16 limit of KeyPointVec.size", n);
  if ((long long)(width + 1) * (height + 1) * 255 > 0x7FFFFFFFll) return sf_fail(c, SF_ERANGE, "image too large for a 32-bit integral image");
  SF_HIP(c, hipSetDevice(c->device));
  int rc = brief_ensure(c);
  if (rc != SF_OK) return rc;
  if ((rc = store_reserve(c, c->store, c->store.slots + 1, n, c->brief_bytes)) != SF_OK) return rc;
  if (out_rows && (rc = sf_buf_reserve(c, c->ex_rows, 16)) != SF_OK) return rc;
  Store& st = c->store;
  const int slot = st.slots;
  if ((rc = sf_launch_extract(c, d_left, width, height, pitch, d_kpts, d_right_x, d_status, n, cam, c->brief_bytes,
                              (const int8_t*)c->brief_tests.p, (uint32_t*)st.desc.p, (float*)st.xyz.p, (float4*)st.kp.p,
                              (int4*)st.meta.p, st.kcap, st.w, slot, d_desc_out, d_xyz_out, d_kpts_out,
                              out_rows ? (int32_t*)c->ex_rows.p : nullptr)) != SF_OK)
    return rc;
  st.slots += 1;
  if (out_slot) *out_slot = slot;
  if (out_rows) {
    SF_HIP(c, hipMemcpyAsync(out_rows, c->ex_rows.p, 4, hipMemcpyDeviceToHost, c->stream));
    SF_HIP(c, hipStreamSynchronize(c->stream));
  }
  return SF_OK;
}

extern "C" void sf_detector_defaults(sf_detector_params* p) {
  if (!p) return;
  p->max_features = 1000;      // Vis/MaxFeatures [upstream rtabmap Parameters.h]; the reference sets only Vis/MinInliers
  p->quality_level = 0.001;    // GFTT/QualityLevel
  p->min_distance = 3.0;       // GFTT/MinDistance
}

// The GetFeatsAndDesc handler in one call on HOST buffers: upload the pair, detect, track, extract, download the
// response.  Everything between the two copies stays on the device; the keyframe is in the store when it returns.
extern "C" int sf_get_features_and_descriptor(sf_handle c, const uint8_t* left, const uint8_t* right, int32_t width,
                                              int32_t height, int32_t pitch, const sf_stereo_camera* cam,
                                              const sf_detector_params* det, const sf_stereo_flow_params* flow,
                                              uint8_t* desc_out, float* xyz_out, sf_keypoint* kpts_out, int32_t cap_rows,
                                              int32_t* rows_out, int32_t* slot_out) {
  if (!c || !cam || !rows_out || cap_rows < 0) return SF_EINVAL;
  *rows_out = 0;
  if (!left || !right || width < 3 || height < 3 || pitch < width)
    return sf_fail(c, SF_EINVAL, "stereo pair missing or malformed (%d x %d, pitch %d)", width, height, pitch);
  sf_detector_params dp;
  if (det) dp = *det; else sf_detector_defaults(&dp);
  if (dp.max_features <= 0 || dp.max_features > SF_MAX_FEATURES)
    return sf_fail(c, SF_ERANGE, "max_features %d outside 1 .. %d (KeyPointVec.size is an int16)", dp.max_features, SF_MAX_FEATURES);
  SF_HIP(c, hipSetDevice(c->device));
  int rc = brief_ensure(c);
  if (rc != SF_OK) return rc;
  const size_t img_bytes = ((size_t)width * height + 255) & ~(size_t)255;
  const int maxf = dp.max_features;
  if ((rc = sf_buf_reserve(c, c->ft_images, 2 * img_bytes)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ft_kpts, (size_t)maxf * sizeof(sf_keypoint))) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ft_flow, (size_t)maxf * 16)) != SF_OK) return rc;
  const size_t row_bytes = (size_t)c->brief_bytes + 12 + sizeof(sf_keypoint);
  if ((rc = sf_buf_reserve(c, c->ft_wire, (size_t)maxf * row_bytes + 64)) != SF_OK) return rc;
  uint8_t* d_left = (uint8_t*)c->ft_images.p;
  uint8_t* d_right = d_left + img_bytes;
  SF_HIP(c, hipMemcpy2DAsync(d_left, width, left, pitch, width, height, hipMemcpyHostToDevice, c->stream));
  SF_HIP(c, hipMemcpy2DAsync(d_right, width, right, pitch, width, height, hipMemcpyHostToDevice, c->stream));
  sf_keypoint* d_kpts = (sf_keypoint*)c->ft_kpts.p;
  int32_t n = 0;
  if ((rc = sf_detect_corners_device(c, d_left, width, height, width, maxf, dp.quality_level, dp.min_distance, d_kpts, maxf, &n)) != SF_OK)
    return rc;
  n = std::min(n, maxf);
  float* d_xy = (float*)c->ft_flow.p;                    // [n][2], then x [n], then status [n]
  float* d_rx = d_xy + 2 * (size_t)maxf;
  uint8_t* d_status = (uint8_t*)(d_rx + maxf);
  if ((rc = sf_stereo_correspondences_device(c, d_left, d_right, width, height, width, d_kpts, n, flow, d_xy, d_status, d_rx, nullptr)) != SF_OK)
    return rc;
  uint8_t* d_desc = (uint8_t*)c->ft_wire.p;
  float* d_xyz = (float*)(d_desc + (((size_t)maxf * c->brief_bytes + 15) & ~(size_t)15));
  sf_keypoint* d_kp_out = (sf_keypoint*)(d_xyz + 3 * (size_t)maxf);
  int32_t slot = -1, rows = 0;
  if ((rc = sf_extract_keyframe_device(c, d_left, width, height, width, d_kpts, d_rx, d_status, n, cam, &slot, &rows, d_desc,
                                       d_xyz, d_kp_out)) != SF_OK)
    return rc;
  *rows_out = rows;
  if (slot_out) *slot_out = slot;
  const int32_t k = std::min(rows, cap_rows);
  if (k > 0) {
    if (desc_out) SF_HIP(c, hipMemcpyAsync(desc_out, d_desc, (size_t)k * c->brief_bytes, hipMemcpyDeviceToHost, c->stream));
    if (xyz_out) SF_HIP(c, hipMemcpyAsync(xyz_out, d_xyz, (size_t)k * 12, hipMemcpyDeviceToHost, c->stream));
    if (kpts_out) SF_HIP(c, hipMemcpyAsync(kpts_out, d_kp_out, (size_t)k * sizeof(sf_keypoint), hipMemcpyDeviceToHost, c->stream));
    SF_HIP(c, hipStreamSynchronize(c->stream));
  }
  return SF_OK;
}

// n keyframes from device images to n store slots in ONE launch sequence: detector, stereo correspondence and
// extraction each run once over the batch (blockIdx = image), the corner counts stay in device memory between them, the
// host never waits.  Same per-keyframe results as sf_get_features_and_descriptor.
extern "C" int sf_get_features_and_descriptor_batch_device(sf_handle c, const uint8_t* d_left, const uint8_t* d_right,
                                                           int32_t n_keyframes, int32_t width, int32_t height, int32_t pitch,
                                                           size_t image_stride, const sf_stereo_camera* cam,
                                                           const sf_detector_params* det, const sf_stereo_flow_params* flow,
                                                           int32_t* first_slot_out, int32_t* d_rows_out, uint8_t* d_desc_out,
                                                           float* d_xyz_out, sf_keypoint* d_kpts_out) {
  if (!c || !cam || n_keyframes < 0) return SF_EINVAL;
  if (n_keyframes == 0) { if (first_slot_out) *first_slot_out = c->store.slots; return SF_OK; }
  if (!d_left || !d_right || width < 3 || height < 3 || pitch < width || image_stride < (size_t)pitch * height)
    return sf_fail(c, SF_EINVAL, "stereo pairs missing or malformed (%d x %d, pitch %d, stride %zu)", width, height, pitch, image_stride);
  sf_detector_params dp;
  if (det) dp = *det; else sf_detector_defaults(&dp);
  if (dp.max_features <= 0 || dp.max_features > SF_MAX_FEATURES)
    return sf_fail(c, SF_ERANGE, "max_features %d outside 1 .. %d (KeyPointVec.size is an int16)", dp.max_features, SF_MAX_FEATURES);
  if (!(dp.quality_level > 0.0) || !(dp.min_distance >= 0.0))
    return sf_fail(c, SF_EINVAL, "qualityLevel must be > 0 and minDistance >= 0 (cv::goodFeaturesToTrack asserts the same)");
  sf_stereo_flow_params prm;
  if (flow) prm = *flow; else sf_stereo_flow_defaults(&prm);
  if (prm.win_width <= 2 || prm.win_height <= 2 || (long long)prm.win_width * prm.win_height > 1024 || prm.max_level < 0 ||
      prm.max_level > 15 || !(prm.epsilon == prm.epsilon))
    return sf_fail(c, SF_EINVAL, "stereo flow parameters out of range (see sf_stereo_correspondences_device)");
  if ((long long)(width + 1) * (height + 1) * 255 > 0x7FFFFFFFll) return sf_fail(c, SF_ERANGE, "image too large for a 32-bit integral image");
  SF_HIP(c, hipSetDevice(c->device));
  int rc = brief_ensure(c);
  if (rc != SF_OK) return rc;
  const int maxf = dp.max_features, n = n_keyframes;
  const size_t rows_all = (size_t)maxf * n;
  if ((rc = sf_buf_reserve(c, c->ft_kpts, rows_all * sizeof(sf_keypoint))) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ft_flow, rows_all * 16)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->ft_counts, (size_t)n * 4)) != SF_OK) return rc;
  if ((rc = store_reserve(c, c->store, c->store.slots + n, maxf, c->brief_bytes)) != SF_OK) return rc;
  sf_keypoint* d_kpts = (sf_keypoint*)c->ft_kpts.p;
  int32_t* d_n = (int32_t*)c->ft_counts.p;
  if ((rc = sf_launch_detect_corners_batch(c, d_left, image_stride, n, width, height, pitch, maxf, dp.quality_level,
                                           dp.min_distance, d_kpts, maxf, d_n)) != SF_OK)
    return rc;
  float* d_xy = (float*)c->ft_flow.p;                    // [n][maxf][2], then x [n][maxf], then status [n][maxf]
  float* d_rx = d_xy + 2 * rows_all;
  uint8_t* d_status = (uint8_t*)(d_rx + rows_all);
  if ((rc = sf_launch_stereo_flow_batch(c, d_left, d_right, image_stride, n, width, height, pitch, d_kpts, maxf, d_n, &prm,
                                        d_xy, d_status, d_rx, nullptr)) != SF_OK)
    return rc;
  Store& st = c->store;
  const int slot = st.slots;
  if ((rc = sf_launch_extract_batch(c, d_left, image_stride, n, width, height, pitch, d_kpts, d_rx, d_status, maxf, d_n, cam,
                                    c->brief_bytes, (const int8_t*)c->brief_tests.p, (uint32_t*)st.desc.p, (float*)st.xyz.p,
                                    (float4*)st.kp.p, (int4*)st.meta.p, st.kcap, st.w, slot, d_desc_out, d_xyz_out,
                                    d_kpts_out, d_rows_out)) != SF_OK)
    return rc;
  st.slots += n;
  if (first_slot_out) *first_slot_out = slot;
  return SF_OK;
}

extern "C" int sf_store_size(sf_handle c, int32_t* n_slots) {
  if (!c || !n_slots) return SF_EINVAL;
  *n_slots = c->store.slots;
  return SF_OK;
}

extern "C" int sf_store_clear(sf_handle c) {
  if (!c) return SF_EINVAL;
  (void)sf_lanes_touch(c, true);
  SF_HIP(c, hipStreamSynchronize(c->stream));
  c->store.slots = 0;
  return SF_OK;
}

// ---- verification pipeline ------------------------------------------------------------------------
// lists: the correspondence lists live in HBM (stage kernels, or the fused kernel with SF_OPT_DEBUG_CORR); the fused
// kernel otherwise keeps them in LDS and the two kcap-entry arrays per pair are not needed
// Bytes of each workspace array for a launch sequence of n pairs: what ws_reserve reserves, as a pure function (also what
// sf_debug_plan_workspace reports, so that a test without a GPU can hold it against what each launch form writes).
struct WsBytes { size_t corr1, corr2, hdr1, hdr2, pass1, pass2, list1, list3, flags; };
static WsBytes ws_bytes(int n, int kcap, bool lists) {
  const size_t np = (size_t)n;
  WsBytes w;
  w.corr1 = w.corr2 = lists ? np * kcap * 4 : 0;
  w.hdr1 = w.hdr2 = np * sizeof(CorrHeader);
  w.pass1 = w.pass2 = np * sizeof(PassState);
  w.list1 = w.list3 = np * 4;
  w.flags = np;
  return w;
}

static int ws_reserve(sf_context* c, int n, int kcap, bool lists) {
  int rc;
  const size_t np = (size_t)n;
  const WsBytes w = ws_bytes(n, kcap, lists);
  if (lists) {
    if ((rc = sf_buf_reserve(c, c->corr1, w.corr1)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->corr2, w.corr2)) != SF_OK) return rc;
  }
  if ((rc = sf_buf_reserve(c, c->hdr1, w.hdr1)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->hdr2, w.hdr2)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->pass1, w.pass1)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->pass2, w.pass2)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->list1, w.list1)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->list3, w.list3)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->flags, w.flags)) != SF_OK) return rc;
#ifndef SF_CHAIN_TRACE
  if (getenv("SF_DIAG")) {     // experiment: eight 64-bit diagnostic counters the kernels may bump (sf_debug_counters)
    // [0..511]: counters; then two 64-bit planes of [pair][kcap] per-point records of the guided pass
    const size_t need = 4096 + 2 * np * (size_t)kcap * 8;
    if (c->trace.bytes < need) {
      if ((rc = sf_buf_reserve(c, c->trace, need)) != SF_OK) return rc;
      SF_HIP(c, hipMemsetAsync(c->trace.p, 0, c->trace.bytes, c->stream));
    }
    c->dparams.dbg_trace = (unsigned long long*)c->trace.p;
  }
#endif
#ifdef SF_CHAIN_TRACE
  if ((rc = sf_buf_reserve(c, c->trace, np * SF_TRACE_SLOTS * 8)) != SF_OK) return rc;
  SF_HIP(c, hipMemsetAsync(c->trace.p, 0, np * SF_TRACE_SLOTS * 8, c->stream));
  c->dparams.dbg_trace = (unsigned long long*)c->trace.p;
#endif
  c->ws_pairs = n;
  c->ws_kcap = kcap;
  return SF_OK;
}

static const int SF_CHUNK = 131072;  // pairs per launch sequence (bounds the workspace: ~4 KiB / pair at K = 500);
                                     // every launch ends with the latency tail of its last surviving pairs, so few, big chunks

// Which form the 3D-3D verification of n pairs takes: the fused kernel (one workgroup carries a pair through matching
// and both motion-estimation chains) or the split form (k_match_split over all pairs + k_chain over the survivors).
// On one stream the fused kernel wins (its chains overlap other pairs' matching inside the launch: 19.5 against 17.1 M
// pairs/s at the bench shape); when sf_step_issue deals the steps over several streams the neighbouring step fills a
// launch's tail anyway and the split form is faster (21.9 against 20.7 M pairs/s; 18.3 against 16.8 M at 40 000
// keyframes): its matching kernel keeps four "to" tiles per wavefront at three workgroups per CU.  Frames that put
// the fused kernel into its WIDE form (K = 1000), 512-bit descriptors and queries of more than 65 536 candidates
// measured equal or slower in the split form and stay fused, and so do small queries (the reference's own cadence of 20
// candidates per tick: one launch instead of three).  SF_FUSED=2 forces the split form everywhere.
static bool sf_use_split(const sf_context* c, const StoreView& v, int n) {
  if (c->split) return sf_split_applicable(c, v);
  // bundle adjustment on: the fused kernel does not apply (the adjustment is a launch of its own, the chain is cut around
  // it); the split form wherever it exists, else the stage kernels
  if (c->dparams.bundle_adjustment && c->dparams.estimation_type == 0 && c->fused) return sf_split_applicable(c, v);
  if (!c->split_auto || !c->in_overlapped_step) return false;
  if (c->dparams.estimation_type != 0 || v.w != 8 || n > 65536 || n < c->split_auto_min || !sf_split_applicable(c, v))
    return false;
  return sf_fused_lds_bytes(c, v) * 4 <= 160 * 1024;       // (not the WIDE form: sf_launch_verify_fused)
}

// ONE object decides the launch form of a verification call AND what its workspace must hold: it is made once per call
// from the call's total pair count, the workspace is reserved from it and every chunk is launched from it.  (Round 3 made
// the form decision per chunk and the reservation per call: the second chunk of a 140 000-candidate step took the split
// form, which writes correspondence lists, on a workspace reserved for the fused form, which has none -- a device
// out-of-bounds write; tests/test_gpu_step.py::test_step_queries_across_the_form_and_chunk_boundaries.)
struct VerifyPlan {
  enum Form { STAGES = 0, FUSED = 1, SPLIT = 2, SPLIT_PNP = 3, HALVES = 4 } form = STAGES;
  bool lists = true;        // the correspondence lists live in HBM (every form but the plain fused kernel)
  bool single = false;      // one launch sequence on one stream: a pair's index IS its position in the call
  bool streams() const { return single && (form == FUSED || form == SPLIT || form == SPLIT_PNP); }   // chain kernels that can
};                                                                                                // stream accepted results

static VerifyPlan verify_plan(const sf_context* c, const StoreView& v, int n) {
  VerifyPlan p;
  if (c->overlap && n >= c->overlap_min_pairs) { p.form = VerifyPlan::HALVES; p.lists = true; p.single = false; return p; }
  p.single = n <= SF_CHUNK;
  if (c->chain_pnp && sf_split_pnp_applicable(c, v)) { p.form = VerifyPlan::SPLIT_PNP; p.lists = true; }
  else if (sf_use_split(c, v, n)) { p.form = VerifyPlan::SPLIT; p.lists = true; }
  else if (sf_fused_lds_bytes(c, v) != 0 && !c->dparams.bundle_adjustment) { p.form = VerifyPlan::FUSED; p.lists = c->debug_corr; }
  else { p.form = VerifyPlan::STAGES; p.lists = true; }
  return p;
}

// What the launches of a form WRITE into the workspace for a sequence of m pairs -- stated here independently of
// ws_bytes / ws_reserve, from the kernels' own indexing (k_verify.hip, k_match.hip, k_ransac.hip, k_pnp.hip,
// k_guided.hip): lists are [pair][kcap] words, headers / states / flags one entry per pair, work lists one int per pair.
// (The `4822be5` fault of round 3 was a form that writes lists on a workspace reserved without them.)
static WsBytes form_writes(VerifyPlan::Form form, int m, int kcap, bool debug_corr, bool ba) {
  const size_t np = (size_t)m, list = np * (size_t)kcap * 4;
  WsBytes w = {};
  switch (form) {
    case VerifyPlan::FUSED:            // lists / headers / states only with SF_OPT_DEBUG_CORR; flags with it too
      if (debug_corr) { w.corr1 = w.corr2 = list; w.hdr1 = w.hdr2 = np * sizeof(CorrHeader); w.pass1 = w.pass2 = np * sizeof(PassState); w.flags = np; }
      break;
    case VerifyPlan::SPLIT:            // k_match_split: corr1, hdr1, pass1, list1 (+ hdr2 / pass2 / flags of non-survivors with
      w.corr1 = list; w.hdr1 = np * sizeof(CorrHeader); w.pass1 = np * sizeof(PassState); w.list1 = np * 4;   // the debug option)
      if (debug_corr || ba) { w.corr2 = list; w.hdr2 = np * sizeof(CorrHeader); w.pass2 = np * sizeof(PassState); w.flags = np; }
      break;
    case VerifyPlan::SPLIT_PNP:        // k_chain_pnp hands everything over through the workspace
    case VerifyPlan::STAGES:
    case VerifyPlan::HALVES:
      w.corr1 = w.corr2 = list; w.hdr1 = w.hdr2 = np * sizeof(CorrHeader); w.pass1 = w.pass2 = np * sizeof(PassState);
      w.list1 = w.list3 = np * 4; w.flags = np;
      break;
  }
  return w;
}

// include/sf_experimental.h: the plan of a verification call and its workspace, computed WITHOUT a device (no HIP call):
// out[0] = form, out[1] = lists, out[2] = single, out[3] = pairs of the largest launch sequence, out[4..12] = bytes
// ws_reserve reserves (corr1, corr2, hdr1, hdr2, pass1, pass2, list1, list3, flags), out[13..21] = bytes the form's
// launches write for that sequence.
extern "C" int sf_debug_plan_workspace(const sf_params* p, int32_t kcap, int32_t desc_words, int32_t n_pairs,
                                       int32_t in_overlapped_step, int32_t debug_corr, int64_t* out, int32_t n_out) {
  if (!p || !out || n_out < 22 || kcap <= 0 || (kcap & 63) || (desc_words != 8 && desc_words != 16 && desc_words != 64 && desc_words != 128) || n_pairs <= 0)
    return SF_EINVAL;
  sf_context* c = new (std::nothrow) sf_context();
  if (!c) return SF_ENOMEM;
  c->params = *p;
  int rc = fill_device_params(c);
  if (rc == SF_OK) {
    if (const char* v = getenv("SF_FUSED")) { c->fused = atoi(v) != 0; c->split = atoi(v) == 2; }
    if (const char* v = getenv("SF_STEP_SPLIT")) c->split_auto = atoi(v) != 0;
    if (const char* v = getenv("SF_CHAIN_PNP")) c->chain_pnp = atoi(v) != 0;
    if (const char* v = getenv("SF_OVERLAP")) c->overlap = atoi(v) != 0;
    c->in_overlapped_step = in_overlapped_step != 0;
    c->debug_corr = debug_corr != 0;
    StoreView v = {};
    v.kcap = kcap; v.w = desc_words; v.n_slots = 1;
    const VerifyPlan plan = verify_plan(c, v, n_pairs);
    const int seq = plan.form == VerifyPlan::HALVES ? std::min((std::min(n_pairs, 2 * SF_CHUNK) + 1) / 2, SF_CHUNK)
                                                    : std::min(n_pairs, SF_CHUNK);
    const bool lists = plan.form == VerifyPlan::HALVES ? true : plan.lists;
    const WsBytes r = ws_bytes(seq, kcap, lists);
    const WsBytes w = form_writes(plan.form == VerifyPlan::HALVES ? VerifyPlan::STAGES : plan.form, seq, kcap, c->debug_corr,
                                  c->dparams.bundle_adjustment != 0);
    out[0] = (int64_t)plan.form; out[1] = plan.lists; out[2] = plan.single; out[3] = seq;
    const size_t rs[9] = {r.corr1, r.corr2, r.hdr1, r.hdr2, r.pass1, r.pass2, r.list1, r.list3, r.flags};
    const size_t ws[9] = {w.corr1, w.corr2, w.hdr1, w.hdr2, w.pass1, w.pass2, w.list1, w.list3, w.flags};
    for (int i = 0; i < 9; ++i) { out[4 + i] = (int64_t)rs[i]; out[13 + i] = (int64_t)ws[i]; }
  }
  delete c;
  return rc;
}

// One launch sequence for m <= SF_CHUNK pairs on ctx's stream and workspace, in the form the call's plan names.
static int verify_sequence(sf_context* ctx, const StoreView& view, const int32_t* d_from, const int32_t* d_to, int m,
                           sf_result* d_out, const VerifyPlan& plan) {
  int rc;
  ctx->dparams.dbg_corr = ctx->debug_corr ? 1 : 0;
  switch (plan.form) {
    case VerifyPlan::SPLIT_PNP:
      ctx->last_lists_valid = true;
      return sf_launch_verify_split(ctx, view, d_from, d_to, m, d_out);
    case VerifyPlan::SPLIT:
      // (pass-2 lists only with the option or the bundle adjustment, whose launches read them; pass-1 lists always)
      ctx->last_lists_valid = ctx->debug_corr || ctx->dparams.bundle_adjustment != 0;
      return sf_launch_verify_split(ctx, view, d_from, d_to, m, d_out);
    case VerifyPlan::FUSED:
      // one launch: every pair's whole two-pass pipeline inside its workgroup (k_verify.hip); no work lists
      ctx->last_lists_valid = ctx->debug_corr;
      return sf_launch_verify_fused(ctx, view, d_from, d_to, m, d_out);
    default: break;
  }
  ctx->last_lists_valid = true;
  SF_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));   // work-list counters of the stage kernels
  if ((rc = sf_launch_match_global(ctx, view, d_from, d_to, m)) != SF_OK) return rc;
  const bool pnp = ctx->dparams.estimation_type == 1;
  if ((rc = (pnp ? sf_launch_pnp : sf_launch_ransac)(ctx, view, d_from, d_to, m, 1)) != SF_OK) return rc;
  if ((rc = sf_launch_guided(ctx, view, d_from, d_to, m)) != SF_OK) return rc;
  if ((rc = (pnp ? sf_launch_pnp : sf_launch_ransac)(ctx, view, d_from, d_to, m, 2)) != SF_OK) return rc;
  return sf_launch_finalize(ctx, m, d_out);
}

// The shadow context of the two-stream path (see sf_context::twin): created on first use.
static int ensure_twin(sf_context* c) {
  if (!c->twin) {
    sf_context* t = new (std::nothrow) sf_context();
    if (!t) return sf_fail(c, SF_ENOMEM, "out of host memory");
    t->device = c->device;
    // (a lowest-priority stream was tried so that the first half would match first and its estimation kernels
    //  overlap the second half's matching: no gain -- both matching kernels still share the chip)
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
      delete t;
      return sf_fail(c, SF_EHIP, "hipStreamCreateWithFlags failed");
    }
    t->own_stream = true;
    c->twin = t;
    int rc = sf_buf_reserve(t, t->counters, 64);
    if (rc != SF_OK) { c->err = t->err; return rc; }
    SF_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    SF_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  sf_context* t = c->twin;
  t->params = c->params;          // parameters may have changed since the last batch
  t->dparams = c->dparams;
  t->match_variant = c->match_variant;
  t->match_mfma = c->match_mfma;
  t->fused = c->fused;
  t->split = c->split;
  t->chain_pnp = c->chain_pnp;
  t->chain_nw = c->chain_nw;
  t->chain_pnp_nw = c->chain_pnp_nw;
  t->ba_nw = c->ba_nw;
  t->ba_occ = c->ba_occ;
  t->debug_corr = c->debug_corr;
  t->prof = c->prof;
  return SF_OK;
}

// d_from / d_to / d_out: device pointers. Asynchronous on the handle's stream.
//
// With SF_OVERLAP=1 batches of at least `overlap_min_pairs` pairs are cut in two halves that run the STAGE
// kernels on two streams, so that the motion-estimation kernels (a few thousand workgroups of latency-bound
// fp64 chains, issue ports idle) of one half can overlap the matching kernel (issue-bound, indifferent to
// 2 / 3 / 4 resident workgroups per CU) of the other.  Measured on the bench step (10 000 pairs): +4-6 % with
// the 3D-3D estimator, +3 % with PnP -- the two matching kernels start together and share the chip, so most
// of the estimation work still runs after both.  Off by default: one launch sequence per chunk keeps the
// per-kernel durations exclusive (what the roofline is computed from) for a gain inside the box-to-box spread.
// The second stream starts after everything already queued on the handle's stream and the handle's stream
// continues only after the second has finished, so the call keeps its single-stream semantics either way.
static int verify_device(sf_context* c, const Store& st, const int32_t* d_from, const int32_t* d_to, int n,
                         sf_result* d_out, const VerifyPlan* given = nullptr) {
  if (n <= 0) return SF_OK;
  if (st.slots <= 0) return sf_fail(c, SF_EINVAL, "keyframe store is empty");
  const StoreView view = sf_store_view(st);
  const VerifyPlan plan = given ? *given : verify_plan(c, view, n);
  int rc;
  if (plan.form == VerifyPlan::HALVES) {
    if ((rc = ensure_twin(c)) != SF_OK) return rc;
    sf_context* t = c->twin;
    VerifyPlan stages;                                   // (two FUSED halves on two streams were measured too and gain nothing)
    const int span = std::min(n, 2 * SF_CHUNK);          // pairs per round: one chunk per stream
    const int half0 = (std::min(span, n) + 1) / 2;
    if ((rc = ws_reserve(c, std::min(half0, SF_CHUNK), st.kcap, stages.lists)) != SF_OK) return rc;
    if ((rc = ws_reserve(t, std::min(half0, SF_CHUNK), st.kcap, stages.lists)) != SF_OK) { c->err = t->err; return rc; }
    SF_HIP(c, hipEventRecord(c->ev_fork, c->stream));
    SF_HIP(c, hipStreamWaitEvent(t->stream, c->ev_fork, 0));
    c->ws_split = 0;
    for (int off = 0; off < n; off += span) {
      const int m = std::min(span, n - off);
      const int ma = (m + 1) / 2, mb = m - ma;
      if (off == 0) c->ws_split = ma;
      if ((rc = verify_sequence(c, view, d_from + off, d_to + off, ma, d_out + off, stages)) != SF_OK) return rc;
      if (mb > 0 && (rc = verify_sequence(t, view, d_from + off + ma, d_to + off + ma, mb, d_out + off + ma, stages)) != SF_OK) {
        c->err = t->err;
        return rc;
      }
    }
    SF_HIP(c, hipEventRecord(c->ev_join, t->stream));
    SF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return SF_OK;
  }
  if ((rc = ws_reserve(c, std::min(n, SF_CHUNK), st.kcap, plan.lists)) != SF_OK) return rc;
  c->ws_split = 0;
  for (int off = 0; off < n; off += SF_CHUNK) {
    const int m = std::min(SF_CHUNK, n - off);
    if ((rc = verify_sequence(c, view, d_from + off, d_to + off, m, d_out + off, plan)) != SF_OK) return rc;
  }
  return SF_OK;
}

extern "C" int sf_verify_pairs_device(sf_handle c, const int32_t* d_from, const int32_t* d_to, int32_t n,
                                      sf_result* d_out) {
  if (!c || n < 0 || (n > 0 && (!d_from || !d_to || !d_out))) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return verify_device(c, c->store, d_from, d_to, n, d_out);
}

extern "C" int sf_verify_matches_device(sf_handle c, const sf_match* matches, int32_t n, int32_t slot_base_other,
                                        int32_t slot_base_local, sf_result* d_out) {
  if (!c || n < 0 || (n > 0 && (!matches || !d_out))) return SF_EINVAL;
  if (n == 0) return SF_OK;
  SF_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = sf_buf_reserve(c, c->pair_from, (size_t)n * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->pair_to, (size_t)n * 4)) != SF_OK) return rc;
  // pinned staging of the two slot lists; the previous call's copies must have left it
  if (c->pairs_staged) SF_HIP(c, hipEventSynchronize(c->pairs_staged));
  const size_t need = (size_t)n * 8;
  if (need > c->pairs_pinned_bytes) {
    if (c->pairs_pinned) (void)hipHostFree(c->pairs_pinned);
    c->pairs_pinned = nullptr;
    c->pairs_pinned_bytes = 0;
    if (hipHostMalloc(&c->pairs_pinned, need + need / 2, hipHostMallocDefault) != hipSuccess)
      return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) failed", need + need / 2);
    c->pairs_pinned_bytes = need + need / 2;
  }
  int32_t* hf = (int32_t*)c->pairs_pinned;
  int32_t* ht = hf + n;
  for (int i = 0; i < n; ++i) {
    hf[i] = slot_base_other + matches[i].idx_other;   // "from" = the querying robot's frame
    ht[i] = slot_base_local + matches[i].idx_local;   // "to"   = the computing robot's frame
    if (hf[i] < 0 || hf[i] >= c->store.slots || ht[i] < 0 || ht[i] >= c->store.slots)
      return sf_fail(c, SF_ERANGE, "match %d: slot (%d,%d) outside the store (%d slots)", i, hf[i], ht[i], c->store.slots);
  }
  SF_HIP(c, hipMemcpyAsync(c->pair_from.p, hf, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  SF_HIP(c, hipMemcpyAsync(c->pair_to.p, ht, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if (!c->pairs_staged) SF_HIP(c, hipEventCreateWithFlags(&c->pairs_staged, hipEventDisableTiming));
  SF_HIP(c, hipEventRecord(c->pairs_staged, c->stream));
  return verify_device(c, c->store, (const int32_t*)c->pair_from.p, (const int32_t*)c->pair_to.p, n, d_out);
}

// Called by the NN filter behind the refinement launch of a prefix level (k_nn.hip): candidate pair list on
// the device, then the verification of every candidate slot, all on the handle's stream.
// Hands the selected accepted-result block to the verification kernels of the launch that follows (fused kernel, chain
// kernels) -- by value, in their kernel arguments: the counter is word 4 of the candidate list's header, zeroed with the
// candidate count before the filter ran (or the caller's own word).
static int arm_accept_stream(sf_context* c, const unsigned* d_count) {
  c->accept_streamed = false;
  if (c->accept_sel < 0 || !c->accept_blocks[c->accept_sel].set) return SF_OK;
  sf_context::AcceptHost& ab = c->accept_blocks[c->accept_sel];
  // Every verified slot may be accepted: a block with fewer record slots than the launch has pairs could lose
  // records (the kernel drops what does not fit and the slot counter is not the caller's to read), so such a block
  // is not armed -- sf_accept_stream_status then reports streamed = 0 and the caller takes the compaction.
  if ((unsigned)ab.s.cap < c->spec.grid) return SF_OK;
  c->dparams.accept = ab.s;
  if (!ab.s.ext_counter) c->dparams.accept.counter = const_cast<unsigned*>(d_count) + 4;
  c->dparams.accept_on = 1;
  c->accept_streamed = true;
  c->accept_armed = true;
  return SF_OK;
}

// The verification of `grid` pair slots taken from a (row, column) list on the device -- the NN filter's candidates
// (speculative path) or the device walk's matches -- with the accepted-result stream armed where the launch form has it.
int sf_spec_launch(sf_context* c, const void* d_cand, const unsigned* d_count) {
  const unsigned grid = c->spec.grid;
  if (c->store.slots <= 0) return sf_fail(c, SF_EINVAL, "keyframe store is empty");
  const StoreView view = sf_store_view(c->store);
  const VerifyPlan plan = verify_plan(c, view, (int)grid);
  int rc = SF_OK;
  if (plan.form == VerifyPlan::FUSED && plan.single) {
    // one chunk on the fused kernel: it derives the pairs from the list itself (one launch and its gap less between
    // the NN stage and the verification)
    c->pair_src.cand = (const uint2*)d_cand;
    c->pair_src.count = d_count;
    c->pair_src.n_l = c->nn_local.n; c->pair_src.n_r = c->nn_recv.n;
    c->pair_src.slot_other = c->spec.slot_other; c->pair_src.slot_local = c->spec.slot_local;
    c->pair_src.n_slots = c->store.slots;
  } else {
    hipLaunchKernelGGL(k_spec_pairs, dim3((grid + 255) / 256), dim3(256), 0, c->stream, (const uint2*)d_cand, d_count, grid,
                       c->nn_local.n, c->nn_recv.n, c->spec.slot_other, c->spec.slot_local, c->store.slots,
                       (int32_t*)c->spec_from.p, (int32_t*)c->spec_to.p);
    SF_HIP(c, hipGetLastError());
  }
  // (one chunk, one stream: a pair's index is its slot in the list -- the fused kernel, the 3D-3D chain kernel of the
  //  split form and the PnP estimator's chain kernel stream their accepted results)
  if (plan.streams()) rc = arm_accept_stream(c, d_count);
  if (rc == SF_OK)
    rc = verify_device(c, c->store, (const int32_t*)c->spec_from.p, (const int32_t*)c->spec_to.p, (int)grid,
                       (sf_result*)c->spec_results.p, &plan);
  c->dparams.accept_on = 0;
  c->pair_src = PairSource();
  return rc;
}

static int place_streams(sf_context* c);

extern "C" int sf_find_matches_and_verify_device(sf_handle c, int32_t slot_base_other, int32_t slot_base_local,
                                                 sf_match* out, int32_t cap, int32_t* n_out, sf_result* d_out) {
  if (!c || !n_out || cap < 0 || (cap > 0 && !out)) return SF_EINVAL;
  *n_out = 0;
  c->last_results = nullptr; c->last_results_index = nullptr; c->last_results_n = 0;
  c->accept_streamed = false;
  c->accept_armed = false;
  if (c->nn_local.n <= 0 || c->nn_recv.n <= 0)
    return sf_fail(c, SF_EINVAL, "empty descriptor database (data_handler.py:308 guards this case)");
  SF_HIP(c, hipSetDevice(c->device));
  const int n_l = c->nn_local.n;
  int rc;
  // Speculate only when the walk may return (almost) every local row -- otherwise verifying every candidate
  // would do far more work than the few matches need -- and on the filter path, which has a candidate list.
  const bool speculate = c->params.nn_precision == 1 && c->store.slots > 0 &&
                         std::min(cap, c->params.netvlad_max_matches_nb) >= n_l && getenv("SF_SPECULATE_OFF") == nullptr;
  if (speculate) {
    if (!c->spec.copy_stream) {
      // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and a
      // second stream that lands on the queue of the handle's stream runs BEHIND the verification instead of
      // beside it (seen as soon as another library -- RCCL -- had created streams of its own: +0.17 ms per step).
      // Streams of another priority level get queues of their own, so unless the process raised the queue budget
      // (bench.py sets GPU_MAX_HW_QUEUES=8, measured slightly better than the priority) this stream is created
      // with the highest priority; its work (exact NN re-evaluation, small copies) is what the host waits for.
      // (round 4, late: the stream's place is measured -- place_streams, further down -- and only if that fails created blind)
      int prio_least = 0, prio_greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
      const char* hwq = getenv("GPU_MAX_HW_QUEUES");
      const int prio = (hwq && atoi(hwq) >= 8) ? 0 : prio_greatest;
      if (!c->placement.tried) (void)place_streams(c);
      if (c->placement.copy) { c->spec.copy_stream = c->placement.copy; c->placement.copy = nullptr; }
      else SF_HIP(c, hipStreamCreateWithPriority(&c->spec.copy_stream, hipStreamNonBlocking, prio));
      SF_HIP(c, hipEventCreateWithFlags(&c->spec.ev_refined, hipEventDisableTiming));
      SF_HIP(c, hipEventCreateWithFlags(&c->spec.ev_copied, hipEventDisableTiming));
      SF_HIP(c, hipEventCreateWithFlags(&c->spec_index_staged, hipEventDisableTiming));
    }
    c->spec.grid = (unsigned)(n_l + n_l / 8 + 256);       // room for rows with more than one candidate
    if ((rc = sf_buf_reserve(c, c->spec_from, (size_t)c->spec.grid * 4)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->spec_to, (size_t)c->spec.grid * 4)) != SF_OK) return rc;
    if ((rc = sf_buf_reserve(c, c->spec_results, (size_t)c->spec.grid * sizeof(sf_result))) != SF_OK) return rc;
    c->spec.slot_other = slot_base_other;
    c->spec.slot_local = slot_base_local;
  }
  c->spec.requested = speculate;
  c->spec.launched = false;
  c->spec.valid = false;
  rc = sf_nn_run(c, out, cap, n_out);
  c->spec.requested = false;
  if (rc != SF_OK) return rc;
  const int n = *n_out;
  if (n == 0) return SF_OK;
  if (!(c->spec.launched && c->spec.valid)) {
    // no speculation, or its candidate set was not the one the matches came from: verify the matches now
    // (a wasted speculative verification, if any, is simply queued in front -- what it streamed into the selected
    //  accepted-result block is not this query's answer: streamed = 0, the block's owner resets it)
    c->accept_streamed = false;
    if (!d_out) {       // the caller only wants sf_last_match_results: an internal block takes the results
      if ((rc = sf_buf_reserve(c, c->results, (size_t)n * sizeof(sf_result))) != SF_OK) return rc;
      d_out = (sf_result*)c->results.p;
    }
    if ((rc = sf_verify_matches_device(c, out, n, slot_base_other, slot_base_local, d_out)) != SF_OK) return rc;
    c->last_results = d_out; c->last_results_index = nullptr; c->last_results_n = n;
    return SF_OK;
  }
  // the matches' results are among the speculative ones: index of each match's candidate, then one gather
  for (int i = 0; i < n; ++i) {
    const int f = slot_base_other + out[i].idx_other, t = slot_base_local + out[i].idx_local;
    if (f < 0 || f >= c->store.slots || t < 0 || t >= c->store.slots)
      return sf_fail(c, SF_ERANGE, "match %d: slot (%d,%d) outside the store (%d slots)", i, f, t, c->store.slots);
  }
  SF_HIP(c, hipEventSynchronize(c->spec_index_staged));   // (never recorded: returns at once) previous upload done
  const size_t need = (size_t)n * 4;
  if (need > c->spec_index_pinned_bytes) {
    if (c->spec_index_pinned) (void)hipHostFree(c->spec_index_pinned);
    c->spec_index_pinned = nullptr;
    c->spec_index_pinned_bytes = 0;
    if (hipHostMalloc(&c->spec_index_pinned, need + need / 2, hipHostMallocDefault) != hipSuccess)
      return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) failed", need + need / 2);
    c->spec_index_pinned_bytes = need + need / 2;
  }
  int32_t* hi = (int32_t*)c->spec_index_pinned;
  for (int i = 0; i < n; ++i) {
    const int ci = c->last_row_cand[out[i].idx_local];
    if (ci < 0 || (unsigned)ci >= c->spec.grid) return sf_fail(c, SF_EHIP, "speculation: match %d has no candidate slot", i);
    hi[i] = ci;
  }
  // The gather reads the index list straight from the pinned host block (40 KB over PCIe inside the kernel): an
  // H2D copy queued on the handle's stream would run AFTER the verification it sits behind -- ~15 us of copy
  // engine latency on the step's critical path for nothing.
  c->last_results = (const sf_result*)c->spec_results.p; c->last_results_index = hi; c->last_results_n = n;
  if (!d_out) return SF_OK;       // no gathered copy wanted: sf_last_match_results + an indexed consumer
  constexpr int PIECES = sizeof(sf_result) / 16;
  hipLaunchKernelGGL(k_spec_gather, dim3(((size_t)n * PIECES + 255) / 256), dim3(256), 0, c->stream,
                     (const sf_result*)c->spec_results.p, (const int32_t*)hi, n, d_out);
  SF_HIP(c, hipGetLastError());
  SF_HIP(c, hipEventRecord(c->spec_index_staged, c->stream));   // the block may be rewritten after this
  return SF_OK;
}

// The two compaction kernels, asynchronous: the number of accepted results is left at d_count (device).
// Ordered compaction in ONE launch: a chunk of 256 candidates per workgroup; every chunk publishes its own number of
// accepted candidates as {epoch, count} and sums the counts of the chunks before it (all chunks are resident -- at
// most 1024 of 256 threads -- and a chunk only ever waits for earlier ones).  The epoch (one per launch, kept by the
// handle) makes stale entries of the previous launch unreadable without a memset in between.  Replaces
// k_compact_count + k_compact_move (kept for batches with more chunks than can be resident at once).
constexpr int COMPACT_CHUNK = 256;        // records per workgroup of k_compact_chain
constexpr int COMPACT_MAX_CHUNKS = 1024;  // all resident at once (256 CUs x 8 workgroups of 256 threads)
// state[0 .. chunks): the chunks' {epoch, own count}; state[chunks]: the launch's arrival word {arrived:16, timed out:16,
// sum:32}, zero between launches (the chunk that arrives last reads the total, publishes it and clears the word).
// cap2: record slots behind acc2 (a mirror smaller than the batch: what does not fit is dropped THERE only and shows in
// *total2, which still carries the full count -- the owner's overflow path)
__global__ void __launch_bounds__(COMPACT_CHUNK)
k_compact_chain(const sf_result* __restrict__ res, const int32_t* __restrict__ index, int n, sf_result* __restrict__ acc,
                uint8_t* __restrict__ flags, unsigned long long* __restrict__ state, unsigned epoch,
                int32_t* __restrict__ total, sf_result* __restrict__ acc2, uint8_t* __restrict__ flags2,
                int32_t* __restrict__ total2, int cap2) {
  __shared__ int wsum[COMPACT_CHUNK / 64];
  __shared__ int s_dst[COMPACT_CHUNK];
  __shared__ int s_src[COMPACT_CHUNK];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * COMPACT_CHUNK;
  const int i = base + tid;
  const int src = i < n ? (index ? index[i] : i) : 0;      // candidate i's record (index: e.g. its speculative slot)
  s_src[tid] = src;
  const bool ok = i < n && res[src].success != 0;
  if (i < n && flags) flags[i] = ok ? 1 : 0;
  if (i < n && flags2) flags2[i] = ok ? 1 : 0;
  const unsigned long long bal = __ballot(ok);
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(bal);
  __syncthreads();
  int woff = 0, own = 0;
#pragma unroll
  for (int w = 0; w < COMPACT_CHUNK / 64; ++w) { woff += (w < wave) ? wsum[w] : 0; own += wsum[w]; }
  // every chunk publishes its OWN count at once (tagged with the launch's epoch) and sums its predecessors' -- two
  // hops however many chunks there are, where a chain of inclusive prefixes costs one per chunk; the word itself is
  // the message, so relaxed agent-scope accesses do and no cache is flushed between XCDs
  if (tid == 0)
    __hip_atomic_store(&state[blockIdx.x], ((unsigned long long)epoch << 32) | (unsigned long long)(unsigned)own,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {
    // (the look-back assumes every earlier chunk gets to run: at most COMPACT_MAX_CHUNKS workgroups, all resident.  It
    //  is bounded all the same -- about a second of polling -- so that a launch whose earlier chunks never start, for
    //  whatever reason, ends with the count -1 (the host reports SF_EHIP) instead of spinning forever)
    unsigned sum = 0;
    bool timed_out = false;
    for (int j = lane; j < (int)blockIdx.x; j += 64) {
      unsigned long long v;
      unsigned spins = 0;
      do {
        v = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == epoch) break;
        __builtin_amdgcn_s_sleep(8);
      } while (++spins < (1u << 22));
      timed_out = timed_out || (unsigned)(v >> 32) != epoch;
      sum += (unsigned)v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const bool any_timeout = __any(timed_out);
    if (lane == 0) {
      s_base = (int)sum;
      // the total is published by whichever chunk ARRIVES last, and a time-out anywhere in the launch makes it -1: a
      // chunk that gave up on a predecessor has moved its records to wrong places whatever the other chunks saw
      const unsigned long long mine = (1ull << 48) | ((unsigned long long)(any_timeout ? 1u : 0u) << 32) |
                                      (unsigned long long)(unsigned)own;
      const unsigned long long old = atomicAdd(&state[gridDim.x], mine);
      if ((unsigned)(old >> 48) + 1u == gridDim.x) {
        const unsigned long long tot = old + mine;
        const int t = ((tot >> 32) & 0xFFFFull) ? -1 : (int)(unsigned)(tot & 0xFFFFFFFFull);
        *total = t;
        if (total2) *total2 = t;
        __hip_atomic_store(&state[gridDim.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  s_dst[tid] = ok ? s_base + woff + before : -1;
  __syncthreads();
  const int m = min(COMPACT_CHUNK, n - base);
  for (int e = tid; e < m * 23; e += COMPACT_CHUNK) {
    const int c = e / 23, piece = e - c * 23;
    const int dst = s_dst[c];
    if (dst >= 0) {
      const uint4 v = reinterpret_cast<const uint4*>(res + s_src[c])[piece];
      reinterpret_cast<uint4*>(acc + dst)[piece] = v;
      if (acc2 && dst < cap2) reinterpret_cast<uint4*>(acc2 + dst)[piece] = v;
    }
  }
}

static int compact_launch(sf_context* c, const sf_result* d_results, int n, sf_result* d_accepted, uint8_t* d_flags,
                          int32_t* d_count, const int32_t* index = nullptr, sf_result* d_accepted2 = nullptr,
                          uint8_t* d_flags2 = nullptr, int32_t* d_count2 = nullptr, int cap2 = 0x7FFFFFFF) {
  const int chunks = (n + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
  int rc;
  if ((rc = sf_buf_reserve(c, c->compact_scratch, (size_t)(chunks + 2) * 8)) != SF_OK) return rc;
  if (chunks <= COMPACT_MAX_CHUNKS) {
    if (c->compact_state_chunks != chunks || c->compact_state_ptr != c->compact_scratch.p) {
      // fresh (or regrown / reallocated) state, or another chunk count (the arrival word sits behind the chunks' words):
      // make every epoch tag invalid and the arrival word zero once
      SF_HIP(c, hipMemsetAsync(c->compact_scratch.p, 0, (size_t)(chunks + 2) * 8, c->stream));
      c->compact_state_chunks = chunks;
      c->compact_state_ptr = c->compact_scratch.p;
    }
    if (++c->compact_epoch == 0) c->compact_epoch = 1;
    hipLaunchKernelGGL(k_compact_chain, dim3(chunks), dim3(COMPACT_CHUNK), 0, c->stream, d_results, index, n, d_accepted, d_flags,
                       (unsigned long long*)c->compact_scratch.p, c->compact_epoch, d_count, d_accepted2, d_flags2, d_count2, cap2);
    SF_HIP(c, hipGetLastError());
    if (index && index == (const int32_t*)c->spec_index_pinned)
      SF_HIP(c, hipEventRecord(c->spec_index_staged, c->stream));   // the pinned index block may be rewritten after this
    return SF_OK;
  }
  if (index || d_accepted2) return sf_fail(c, SF_ERANGE, "indexed / mirrored compaction of %d records: more than %d chunks", n, COMPACT_MAX_CHUNKS);
  c->compact_state_chunks = 0;                  // (the two-kernel form reuses the buffer as plain counts)
  int32_t* d_chunk = (int32_t*)c->compact_scratch.p;
  const int chunks2 = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_compact_count, dim3(chunks2), dim3(1024), 0, c->stream, d_results, n, d_flags, d_chunk);
  hipLaunchKernelGGL(k_compact_move, dim3(chunks2), dim3(1024), 0, c->stream, d_results, n, d_accepted,
                     (const int32_t*)d_chunk, d_count);
  SF_HIP(c, hipGetLastError());
  return SF_OK;
}

extern "C" int sf_compact_accepted_device_async(sf_handle c, const sf_result* d_results, int32_t n,
                                                sf_result* d_accepted, uint8_t* d_flags, int32_t* d_n_accepted) {
  if (!c || n < 0 || !d_n_accepted || (n > 0 && (!d_results || !d_accepted))) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  if (n == 0) {
    SF_HIP(c, hipMemsetAsync(d_n_accepted, 0, 4, c->stream));
    return SF_OK;
  }
  return compact_launch(c, d_results, n, d_accepted, d_flags, d_n_accepted);
}

extern "C" int sf_accept_stream_set(sf_handle c, int32_t which, sf_result* records, int32_t* index, uint8_t* flags,
                                    int32_t cap, sf_result* d_records2, uint32_t* d_counter) {
  if (!c || which < 0 || which > 1) return SF_EINVAL;
  if (!records || !index || cap < 1) { c->accept_blocks[which] = sf_context::AcceptHost(); return SF_OK; }   // (unregister)
  sf_context::AcceptHost& ab = c->accept_blocks[which];
  ab.s.records = records; ab.s.index = index; ab.s.flags = flags; ab.s.cap = cap;
  ab.s.records2 = d_records2;
  ab.s.counter = d_counter; ab.s.ext_counter = d_counter ? 1 : 0;
  ab.set = true;
  return SF_OK;
}

extern "C" int sf_accept_stream_select(sf_handle c, int32_t which) {
  if (!c || which < -1 || which > 1) return SF_EINVAL;
  c->accept_sel = which;
  return SF_OK;
}

extern "C" int sf_accept_stream_status(sf_handle c, int32_t* streamed, int32_t* pairs) {
  if (!c || !streamed) return SF_EINVAL;
  *streamed = c->accept_streamed ? 1 : 0;
  if (pairs) *pairs = c->accept_streamed ? (int32_t)c->spec.grid : 0;
  return SF_OK;
}

// ---- the caller's loop body as a begin / retire pair (find_separators.py:59-133) -------------------------------------
// sf_step_issue = s_find_matches_query + the estimate_transformation calls of every returned candidate, QUEUED;
// sf_step_retire = the per-candidate outcome the loop forwards (find_separators.py:97-133).
//
// Round 4: the step is device-resident.  sf_step_issue queues, on ONE stream and with no host wait,
//     NN filter (or fp32 ranking) -> exact re-evaluation -> per-row minima -> argsort + walk (k_walk_*: data_handler.py:
//     191-205) -> verification of the walk's matches, taken from the device list -> accepted separators streaming
//     into the step's host-pinned block,
// and returns; up to `step_depth` steps are in flight, dealt over `step_lanes` streams, and sf_step_retire is the only
// wait.  (Round 3 waited inside sf_step_issue for the row minima and walked them on the host, with the verification of
// EVERY filter candidate running speculatively beside it: any host hiccup landed in the step time -- one 6.9 ms step of
// 20 halved the driver's figure.)  What the device cannot decide -- a candidate set denser than the filter level the handle
// last settled on allows, which takes the prefix ladder of nn_run_filter -- is reported through the pinned status word; the
// retire then runs the query again on the synchronous path below (step_issue_sync: round 3's body), once, and the ladder
// level it settles on serves the following steps.
static int step_block_reserve(sf_context* c, sf_context::StepBlock& b, int32_t cap) {
  if (!b.done) SF_HIP(c, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
  if (cap <= b.cap) return SF_OK;
  if (b.pinned) (void)hipHostFree(b.pinned);
  b.pinned = nullptr; b.cap = 0; b.pinned_bytes = 0;
  const int32_t want = cap + cap / 4 + 64;
  const size_t rec_bytes = (size_t)want * sizeof(sf_result);
  const size_t idx_off = rec_bytes, flag_off = idx_off + (size_t)want * 4, cnt_off = (flag_off + (size_t)want + 63) & ~(size_t)63;
  const size_t match_off = cnt_off + 64, slot_off = match_off + (size_t)want * sizeof(sf_match);
  const size_t word_off = (slot_off + (size_t)want * 4 + 63) & ~(size_t)63;
  const size_t total = word_off + 64;
  if (hipHostMalloc(&b.pinned, total, hipHostMallocDefault) != hipSuccess)
    return sf_fail(c, SF_ENOMEM, "hipHostMalloc(%zu) failed", total);
  b.pinned_bytes = total;
  b.records = (sf_result*)b.pinned;
  b.index = (int32_t*)((char*)b.pinned + idx_off);
  b.flags = (uint8_t*)b.pinned + flag_off;
  b.count = (int32_t*)((char*)b.pinned + cnt_off);
  b.walk_matches = (sf_match*)((char*)b.pinned + match_off);
  b.walk_slots = (int32_t*)((char*)b.pinned + slot_off);
  b.walk_n = (int32_t*)((char*)b.pinned + word_off);
  b.walk_status = b.walk_n + 1;
  b.cap = want;
  for (int32_t i = 0; i < want; ++i) b.index[i] = -1;
  memset(b.flags, 0, (size_t)want);
  *b.count = 0;
  *b.walk_n = 0;
  *b.walk_status = 0;
  int rc = sf_buf_reserve(c, b.dev, 64 + (size_t)want * 8);
  if (rc != SF_OK) return rc;
  return sf_buf_reserve(c, b.dev_records, (size_t)want * sizeof(sf_result));
}

static inline int step_ring(const sf_context* c) { return c->step_depth + 1; }

static int step_settle_all(sf_context* c);

extern "C" int sf_step_mirror_pair(sf_handle c, sf_result* d_records_even, uint32_t* d_counter_even,
                                   sf_result* d_records_odd, uint32_t* d_counter_odd, int32_t cap) {
  if (!c || cap < 0 || ((d_records_even == nullptr) != (d_counter_even == nullptr)) ||
      ((d_records_odd == nullptr) != (d_counter_odd == nullptr)) || ((d_records_even == nullptr) != (d_records_odd == nullptr)))
    return SF_EINVAL;
  if (c->step_inflight) return sf_fail(c, SF_EINVAL, "sf_step_mirror: %d step(s) in flight, retire them first", c->step_inflight);
  c->step_mirror_records[0] = d_records_even;
  c->step_mirror_counter[0] = d_counter_even;
  c->step_mirror_records[1] = d_records_odd;
  c->step_mirror_counter[1] = d_counter_odd;
  c->step_mirror_cap = d_records_even ? cap : 0;
  c->step_mirror_lanes = false;
  c->step_seq = 0;                                                      // the next step is an even one
  return SF_OK;
}

extern "C" int sf_step_mirror(sf_handle c, sf_result* d_records2, uint32_t* d_counter, int32_t cap) {
  return sf_step_mirror_pair(c, d_records2, d_counter, d_records2, d_counter, cap);
}

// ---- SF_OPT_STEP_OVERLAP: the further lanes of the step pipeline ----------------------------------------------------
static void lane_swap(sf_context* c, int k) {
  sf_context::StepLane& L = c->lanes[k - 1];
  std::swap(c->stream, L.stream);
  std::swap(c->aux, L.aux); std::swap(c->ev_filter, L.ev_filter); std::swap(c->ev_walk, L.ev_walk);
#define SF_SWAP(m) std::swap(c->m, L.m)
  SF_SWAP(pair_from); SF_SWAP(pair_to); SF_SWAP(corr1); SF_SWAP(corr2); SF_SWAP(hdr1); SF_SWAP(hdr2); SF_SWAP(pass1);
  SF_SWAP(pass2); SF_SWAP(pass_back); SF_SWAP(dir_mask); SF_SWAP(list1); SF_SWAP(list3); SF_SWAP(counters);
  SF_SWAP(results); SF_SWAP(flags); SF_SWAP(nn_cand); SF_SWAP(spec_from); SF_SWAP(spec_to); SF_SWAP(spec_results);
  SF_SWAP(spec_index); SF_SWAP(compact_scratch); SF_SWAP(step_nn); SF_SWAP(walk_scratch); SF_SWAP(ws_pairs); SF_SWAP(ws_kcap);
  SF_SWAP(nn_count_idx); SF_SWAP(nn_count_primed); SF_SWAP(compact_epoch); SF_SWAP(compact_state_chunks);
  SF_SWAP(compact_state_ptr);
#undef SF_SWAP
}

// A database (or a mask) is about to change: every step in flight is settled first -- waited for, and re-run on the
// synchronous path if its device walk asked for that -- so that no queued kernel reads what the caller is about to
// write and a fallback still sees the state its step was issued on.  `drain` also waits for the lanes' streams.
int sf_lanes_touch(sf_context* c, bool drain) {
  c->db_epoch += 1;
  const int rc = step_settle_all(c);      // (a step whose fallback re-run failed: the error is the caller's to see)
  if (drain)
    for (auto& L : c->lanes)
      if (L.stream) SF_HIP(c, hipStreamSynchronize(L.stream));
  return rc;
}

static int stream_with_own_queue(sf_context* c, hipStream_t* out, bool aux);

static int lane_create(sf_context* c, int k) {
  sf_context::StepLane& L = c->lanes[k - 1];
  if (L.stream) return SF_OK;
  int rc0 = stream_with_own_queue(c, &L.stream, false);
  if (rc0 != SF_OK) return rc0;
  SF_HIP(c, hipEventCreateWithFlags(&L.ev_main, hipEventDisableTiming));
  return sf_buf_reserve(c, L.counters, 64);          // (the work-list counters of the stage kernels; the handle's own
}                                                    //  are reserved at sf_create)

static int lane_enter(sf_context* c, int k) {
  int rc0 = lane_create(c, k);
  if (rc0 != SF_OK) return rc0;
  sf_context::StepLane& L = c->lanes[k - 1];
  if (L.seen_db_epoch != c->db_epoch) {
    // the databases were written through the handle's stream since this lane last looked: wait for that work once
    SF_HIP(c, hipEventRecord(L.ev_main, c->stream));
    SF_HIP(c, hipStreamWaitEvent(L.stream, L.ev_main, 0));
    L.seen_db_epoch = c->db_epoch;
  }
  lane_swap(c, k);
  return SF_OK;
}

extern "C" int sf_step_mirror_streams(sf_handle c, void** stream_even, void** stream_odd) {
  if (!c || !stream_even || !stream_odd) return SF_EINVAL;
  if (c->step_inflight) return sf_fail(c, SF_EINVAL, "sf_step_mirror_streams: %d step(s) in flight, retire them first", c->step_inflight);
  if (!c->step_mirror_records[0] || c->step_mirror_records[0] == c->step_mirror_records[1])
    return sf_fail(c, SF_EINVAL, "sf_step_mirror_streams: needs two distinct mirrors (sf_step_mirror_pair)");
  SF_HIP(c, hipSetDevice(c->device));
  *stream_even = *stream_odd = (void*)c->stream;
  if (c->step_overlap && !c->overlap && c->step_lanes >= 2) {
    c->cur_lane = 1;
    int rc = lane_create(c, 1);
    c->cur_lane = 0;
    if (rc != SF_OK) return rc;
    *stream_odd = (void*)c->lanes[0].stream;
    c->step_mirror_lanes = true;
  }
  return SF_OK;
}

static void step_accept_block(sf_context* c, sf_context::StepBlock& b, sf_result** mirror_rec, uint32_t** mirror_cnt,
                              int32_t* mirror_cap);

// ---- the synchronous body (round 3's step): the NN stage is walked on the host inside the call -----------------------
// Used when the device walk does not apply (SF_OPT_STEP_DEVICE_WALK off, the two-halves verification) and as the fallback
// of a device step whose candidate set was too dense for the filter level.
static int step_issue_sync(sf_context* c, sf_context::StepBlock& b, int32_t slot_base_other, int32_t slot_base_local) {
  const int n_l = c->nn_local.n;
  int rc;
  sf_result* mirror_rec; uint32_t* mirror_cnt; int32_t mirror_cap;
  step_accept_block(c, b, &mirror_rec, &mirror_cnt, &mirror_cap);
  b.matches.resize((size_t)std::max(n_l, 1));
  const int sel_before = c->accept_sel;
  c->accept_sel = 2;
  int32_t n = 0;
  rc = sf_find_matches_and_verify_device(c, slot_base_other, slot_base_local, b.matches.data(), n_l, &n, nullptr);
  c->accept_sel = sel_before;
  b.device_walk = false;
  b.armed = c->accept_armed;                  // the block may hold streamed records (also of an abandoned speculation)
  if (rc != SF_OK) return rc;
  b.n = n;
  b.streamed = c->accept_streamed && n > 0;
  b.pairs = b.armed ? (int32_t)c->spec.grid : 0;
  if (b.streamed) {
    b.slot_of_match.resize((size_t)n);
    const int32_t* ix = c->last_results_index;
    for (int i = 0; i < n; ++i) b.slot_of_match[i] = ix ? ix[i] : i;
  } else if (n > 0) {
    // not streamed (no speculation for this query, or a launch shape the stream does not cover): the accepted results of
    // the matches are compacted, in match order, straight into the block (mirror writes beyond its capacity are dropped
    // and show in the mirror's count: the exchange's overflow path)
    if (n > b.cap) return sf_fail(c, SF_ERANGE, "sf_step_issue: %d matches exceed the block's %d records", n, b.cap);
    if ((rc = compact_launch(c, c->last_results, n, b.records, b.flags, b.count, c->last_results_index,
                             mirror_rec, nullptr, (int32_t*)mirror_cnt, mirror_cap)) != SF_OK) return rc;
  }
  SF_HIP(c, hipEventRecord(b.done, c->stream));
  return SF_OK;
}

// ---- the device-resident bodies --------------------------------------------------------------------------------------
// ---- where the step pipeline's streams sit on the hardware ----------------------------------------------------------
// The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 per priority level by default) and
// the queues onto the FOUR dispatch pipes of the command processor.  A launch whose workgroups do not all fit on the
// chip (every verification launch of a batch step) keeps its pipe's dispatcher busy until the last workgroup is placed:
// a launch on another queue of the SAME pipe waits for that, one on another pipe starts at once
// (tools/ubench/pipe_probe.hip, profiles/r04v_placement/pipe_probe_*.txt: 0.93-1.01 of the blocking launch's duration against
// 0.07).  Which pipe a new stream lands on depends on every stream the process created before -- torch's, RCCL's, the
// caller's -- so the same library ran a step in 0.44 ms or 0.50-0.56 ms depending on whether ONE other stream had been
// used first (profiles/r04v_placement, run k).  Hence: measure.  Twelve candidate streams (six per priority level) are sorted into
// classes by "a long launch on X delays a one-wavefront launch on Y"; the lanes' main streams are taken from classes
// other than the handle's own stream's (and each other's), the second streams -- the nine small dependent launches of
// the device walk, which must never sit behind a verification's dispatch -- from a class no main stream uses,
// highest priority first and on different queues where the class has several.  ~10-20 ms, once per handle.
__global__ void __launch_bounds__(512) k_place_hog(int* sink, int spins) {
  __shared__ int pad[16384];                           // 64 KB: two workgroups per CU, far fewer than the grid holds
  pad[threadIdx.x] = (int)threadIdx.x;
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
  __syncthreads();
  if (pad[(threadIdx.x + 1) & 511] == -1) sink[0] = 1;
}
__global__ void k_place_probe(int* sink) {
  if (threadIdx.x == 999) sink[1] = 1;
}

namespace {
struct PlaceProbe {
  int* d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  int tests = 0;
  // (end of a one-wavefront launch on y - start of a chip-filling launch on x) / duration of the latter; < 0: failed
  float ratio(hipStream_t x, hipStream_t y) {
    ++tests;
    if (hipEventRecord(e0, x) != hipSuccess) return -1.f;
    hipLaunchKernelGGL(k_place_hog, dim3(4096), dim3(512), 0, x, d, 5);
    if (hipEventRecord(e1, x) != hipSuccess) return -1.f;
    hipLaunchKernelGGL(k_place_probe, dim3(1), dim3(64), 0, y, d);
    if (hipEventRecord(e2, y) != hipSuccess) return -1.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventSynchronize(e2) != hipSuccess) return -1.f;
    float t_h = 0.f, t_y = 0.f;
    if (hipEventElapsedTime(&t_h, e0, e1) != hipSuccess || t_h <= 0.f) return -1.f;
    if (hipEventElapsedTime(&t_y, e0, e2) != hipSuccess) return 0.f;      // (the probe ran before the other queue started)
    return std::max(0.f, t_y / t_h);
  }
};
}  // namespace

static int place_streams(sf_context* c) {
  sf_context::StreamPlacement& P = c->placement;
  if (P.tried) return SF_OK;
  P.tried = true;
  if (const char* v = getenv("SF_STREAM_PLACEMENT")) if (atoi(v) == 0) return SF_OK;
  const auto t_begin = std::chrono::steady_clock::now();
  constexpr int NC = 12, MAXCLS = 8;
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  hipStream_t S[NC + 1] = {};
  bool high[NC + 1] = {};
  int cls[NC + 1], queue_of[NC + 1];
  PlaceProbe pr;
  auto cleanup = [&](int rc) {
    for (int i = 1; i <= NC; ++i) if (S[i]) { (void)hipStreamSynchronize(S[i]); (void)hipStreamDestroy(S[i]); }
    if (pr.e0) (void)hipEventDestroy(pr.e0);
    if (pr.e1) (void)hipEventDestroy(pr.e1);
    if (pr.e2) (void)hipEventDestroy(pr.e2);
    if (pr.d) (void)hipFree(pr.d);
    return rc;
  };
  S[0] = c->stream;
  for (int i = 1; i <= NC; ++i) {
    high[i] = (i & 1) == 0;
    if (hipStreamCreateWithPriority(&S[i], hipStreamNonBlocking, high[i] ? prio_greatest : 0) != hipSuccess) { S[i] = nullptr; return cleanup(SF_OK); }
  }
  if (hipMalloc((void**)&pr.d, 64) != hipSuccess || hipEventCreate(&pr.e0) != hipSuccess || hipEventCreate(&pr.e1) != hipSuccess ||
      hipEventCreate(&pr.e2) != hipSuccess) return cleanup(SF_OK);
  // every stream's hardware queue exists before anything is measured (the runtime creates it at the stream's first use).
  // Only THIS handle's streams are waited for (rounds 3-4 waited for the whole device, which also stalled on the work of
  // every other stream of the process -- RCCL's, torch's): work of other streams or processes that runs during the
  // measurement reads as "blocked" and is what the once-more rule below and the abandon path are for; a host that wants the
  // measurement at a quiet moment calls sf_streams_prepare (include/sf_experimental.h) when it has one.
  (void)hipStreamSynchronize(c->stream);
  for (auto& L : c->lanes) if (L.stream) (void)hipStreamSynchronize(L.stream);
  for (int i = 1; i <= NC; ++i) {
    hipLaunchKernelGGL(k_place_probe, dim3(1), dim3(64), 0, S[i], pr.d);
    (void)hipStreamSynchronize(S[i]);
  }
  if (const char* v = getenv("SF_STREAM_PLACEMENT")) if (atoi(v) >= 3) {
    fprintf(stderr, "sepfinder: placement matrix (row: chip-filling launch on X; column: small launch on Y; 0 = the handle's stream, even = highest priority)\n      ");
    for (int y = 0; y <= NC; ++y) fprintf(stderr, " %c%-4d", high[y] ? 'H' : 'n', y);
    fprintf(stderr, "\n");
    for (int x = 0; x <= NC; ++x) {
      fprintf(stderr, "%c%-4d ", high[x] ? 'H' : 'n', x);
      for (int y = 0; y <= NC; ++y) { if (x == y) fprintf(stderr, "    - "); else fprintf(stderr, " %5.2f", pr.ratio(S[x], S[y])); }
      fprintf(stderr, "\n");
    }
  }
  // classes: streams a chip-filling launch on one of which delays the others (same pipe, or same queue)
  int rep[MAXCLS], n_cls = 0;
  bool failed = false;
  for (int i = 0; i <= NC && !failed; ++i) {
    cls[i] = -1;
    for (int k = 0; k < n_cls && cls[i] < 0; ++k) {
      float r = pr.ratio(S[rep[k]], S[i]);
      if (r > 0.3f && r < 0.6f) r = pr.ratio(S[rep[k]], S[i]);      // (something else ran in between: once more)
      if (r < 0.f) { failed = true; break; }
      if (r >= 0.5f) cls[i] = k;
    }
    if (cls[i] < 0 && !failed) {
      if (n_cls == MAXCLS) { failed = true; break; }
      rep[n_cls] = i; cls[i] = n_cls++;
    }
  }
  if (failed || n_cls < 2) {
    snprintf(P.report, sizeof(P.report), "placement: measurement %s (%d classes): streams as created", failed ? "failed" : "found one class", n_cls);
    return cleanup(SF_OK);
  }
  const int lanes = std::min(std::max(c->step_lanes, 1), SF_STEP_MAX_LANES);
  bool cls_main[MAXCLS] = {};
  cls_main[cls[0]] = true;
  bool taken[NC + 1] = {};
  auto take = [&](int i) { hipStream_t s = S[i]; S[i] = nullptr; taken[i] = true; return s; };
  // main streams of lanes 1..: a class no earlier main stream uses, default priority where the class offers it
  int main_cls[SF_STEP_MAX_LANES] = {}; main_cls[0] = cls[0];
  for (int k = 1; k < lanes; ++k) {
    int best = -1;
    for (int pass = 0; pass < 2 && best < 0; ++pass)
      for (int i = 1; i <= NC && best < 0; ++i)
        if (!taken[i] && !cls_main[cls[i]] && (pass == 1 || !high[i])) best = i;
    if (best < 0) break;                       // (fewer classes than lanes: the remaining lanes get streams as before)
    main_cls[k] = cls[best];
    cls_main[cls[best]] = true;
    P.main[k] = take(best);
  }
  // second streams: a class without a main stream -- the one with most candidates -- else the least bad: the fullest class
  int cnt[MAXCLS] = {}, aux_cls = -1;
  for (int i = 1; i <= NC; ++i) if (!taken[i]) cnt[cls[i]] += 1;
  for (int k = 0; k < n_cls; ++k) if (!cls_main[k] && cnt[k] > 0 && (aux_cls < 0 || cnt[k] > cnt[aux_cls])) aux_cls = k;
  const bool aux_free_pipe = aux_cls >= 0;
  if (aux_cls < 0) for (int k = 0; k < n_cls; ++k) if (cnt[k] > 0 && (aux_cls < 0 || cnt[k] > cnt[aux_cls])) aux_cls = k;
  int n_queues = 0;
  if (aux_cls >= 0) {
    // queues inside the class: a launch on the SAME queue ends behind the blocking one (ratio >= 1), one on another queue
    // of the pipe starts when the last workgroup has been placed (ratio ~ 1 - 1 / generations)
    int members[NC], n_m = 0, qrep[NC];
    for (int pass = 0; pass < 2; ++pass)       // highest priority first
      for (int i = 1; i <= NC; ++i) if (!taken[i] && cls[i] == aux_cls && high[i] == (pass == 0)) members[n_m++] = i;
    for (int m = 0; m < n_m; ++m) {
      const int i = members[m];
      queue_of[i] = -1;
      for (int q = 0; q < n_queues && queue_of[i] < 0; ++q) {
        const float r = pr.ratio(S[qrep[q]], S[i]);
        if (r >= 0.985f) queue_of[i] = q;
      }
      if (queue_of[i] < 0) { qrep[n_queues] = i; queue_of[i] = n_queues++; }
    }
    // deal them out queue by queue: lanes' second streams first, then the synchronous call's
    hipStream_t* want[SF_STEP_MAX_LANES + 1];
    int n_w = 0;
    for (int k = 0; k < lanes; ++k) want[n_w++] = &P.aux[k];
    want[n_w++] = &P.copy;
    int w = 0;
    for (int round = 0; round < n_m && w < n_w; ++round)
      for (int q = 0; q < n_queues && w < n_w; ++q)
        for (int m = 0; m < n_m; ++m)
          if (queue_of[members[m]] == q && !taken[members[m]]) { *want[w++] = take(members[m]); break; }
  }
  P.done = true;
  const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  int off = snprintf(P.report, sizeof(P.report), "placement: %d classes, %d tests, %.1f ms; main classes", n_cls, pr.tests, ms);
  for (int k = 0; k < lanes && off < (int)sizeof(P.report) - 8; ++k)
    off += snprintf(P.report + off, sizeof(P.report) - off, " %d%s", main_cls[k], (k == 0 || P.main[k]) ? "" : "?");
  int n_aux = 0;
  for (int k = 0; k < SF_STEP_MAX_LANES; ++k) n_aux += P.aux[k] != nullptr;
  if (off < (int)sizeof(P.report) - 8)
    snprintf(P.report + off, sizeof(P.report) - off, "; second streams: class %d (%s), %d stream(s) on %d queue(s)%s", aux_cls,
             aux_free_pipe ? "no main stream on it" : "SHARED with a main stream", n_aux, n_queues, P.copy ? " + 1 for the synchronous call" : "");
  if (const char* v = getenv("SF_STREAM_PLACEMENT")) if (atoi(v) >= 2) fprintf(stderr, "sepfinder: %s\n", P.report);
  return cleanup(SF_OK);
}

// Runs the stream placement measurement NOW (idempotent; otherwise it runs inside the first step that needs a second
// stream): 60-100 ms, ~100 short chip-filling launches on the handle's stream and on twelve streams of the library's own.
extern "C" int sf_streams_prepare(sf_handle c) {
  if (!c) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return place_streams(c);
}

extern "C" int sf_stream_placement(sf_handle c, char* buf, size_t n) {
  if (!c || !buf || n == 0) return SF_EINVAL;
  snprintf(buf, n, "%s", c->placement.tried ? (c->placement.report[0] ? c->placement.report : "placement: off") : "placement: not measured yet");
  return SF_OK;
}

static int stream_with_own_queue(sf_context* c, hipStream_t* out, bool aux) {
  // a measured place first (place_streams)
  if (!c->placement.tried) (void)place_streams(c);
  if (c->placement.done) {
    hipStream_t& slot = aux ? c->placement.aux[c->cur_lane] : c->placement.main[c->cur_lane];
    if (slot) { *out = slot; slot = nullptr; return SF_OK; }
  }

  // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and a stream that lands
  // on the queue of another runs BEHIND it, not beside it.  Streams of another priority level draw from queues of their
  // own, so unless the process raised the queue budget the library's extra streams get the highest priority.
  // The second stream of a speculative step (exact re-evaluation, row minima, walk: a chain of nine small dependent
  // launches) ALWAYS gets it: beside a matching launch, which holds every register of every CU, a launch of default
  // priority waits for hundreds of microseconds for its first workgroup slot (k_walk_tile_sort: 435 us instead of 32,
  // profiles/r04d_timeline_*), and the step is not done before its walk is.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const char* hwq = getenv("GPU_MAX_HW_QUEUES");
  int prio = (hwq && atoi(hwq) >= 8 && !aux) ? 0 : prio_greatest;
  if (const char* v = getenv(aux ? "SF_AUX_PRIO" : "SF_LANE_PRIO")) prio = atoi(v) > 0 ? prio_greatest : (atoi(v) < 0 ? prio_least : 0);
  SF_HIP(c, hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio));
  return SF_OK;
}

static void step_accept_block(sf_context* c, sf_context::StepBlock& b, sf_result** mirror_rec, uint32_t** mirror_cnt,
                              int32_t* mirror_cap) {
  *mirror_rec = c->step_mirror_records[b.parity];
  *mirror_cnt = c->step_mirror_counter[b.parity];
  *mirror_cap = *mirror_rec ? c->step_mirror_cap : b.cap;
  // no caller mirror: every accepted record also lands in the block's own device buffer (sf_step_result.d_records), so
  // that a multi-GPU host can hand a RETIRED step's separators to its collective without tying buffers to steps in flight
  if (!*mirror_rec) *mirror_rec = (sf_result*)b.dev_records.p;
  sf_context::AcceptHost& ab = c->accept_blocks[2];
  ab.s.records = b.records; ab.s.index = b.index; ab.s.flags = nullptr; ab.s.cap = std::min(b.cap, *mirror_cap);
  ab.s.records2 = *mirror_rec; ab.s.counter = *mirror_cnt; ab.s.ext_counter = *mirror_cnt ? 1 : 0;
  ab.set = true;
}

// Batch mode (the walk may return every local row) on the fp16 filter: the verification of EVERY filter candidate goes
// onto the step's stream straight behind the filter, and the exact re-evaluation, the row minima and the walk run on a
// second stream beside it -- off the chain of dependent launches that decides how soon the lane is free for its next
// step.  The walk's matches then name their candidate's verification slot (walk_slots); round 3 did the same with the
// host in the middle.  A step of the reference's cadence (20 matches of 10 000 rows) would verify 500 x too much this way
// and takes step_issue_serial.
static int step_issue_speculative(sf_context* c, sf_context::StepBlock& b, int32_t slot_base_other, int32_t slot_base_local,
                                  int lim) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n;
  int rc;
  if (!c->aux) {
    if ((rc = stream_with_own_queue(c, &c->aux, true)) != SF_OK) return rc;
    SF_HIP(c, hipEventCreateWithFlags(&c->ev_filter, hipEventDisableTiming));
    SF_HIP(c, hipEventCreateWithFlags(&c->ev_walk, hipEventDisableTiming));
  }
  const size_t min_b = ((size_t)n_l * 8 + 63) & ~(size_t)63, i32_b = ((size_t)n_l * 4 + 63) & ~(size_t)63;
  if ((rc = sf_buf_reserve(c, c->step_nn, 2 * min_b + 2 * i32_b + 64)) != SF_OK) return rc;
  char* base = (char*)c->step_nn.p;
  double* d_min = (double*)base;
  int32_t* d_arg = (int32_t*)(base + min_b);
  int32_t* d_cand = (int32_t*)(base + min_b + i32_b);
  unsigned long long* d_arg64 = (unsigned long long*)(base + min_b + 2 * i32_b);
  int32_t* d_status = (int32_t*)(base + 2 * min_b + 2 * i32_b);
  NnFilterOut fo;
  if ((rc = sf_nn_filter_dev(c, &fo)) != SF_OK) return rc;
  SF_HIP(c, hipEventRecord(c->ev_filter, c->stream));
  // the step's stream: every candidate slot verified (slots past the device-side count are void)
  c->spec.grid = (unsigned)(n_l + n_l / 8 + 256);       // room for rows with more than one candidate
  c->spec.slot_other = slot_base_other;
  c->spec.slot_local = slot_base_local;
  if ((rc = sf_buf_reserve(c, c->spec_from, (size_t)c->spec.grid * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->spec_to, (size_t)c->spec.grid * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->spec_results, (size_t)c->spec.grid * sizeof(sf_result))) != SF_OK) return rc;
  sf_result* mirror_rec; uint32_t* mirror_cnt; int32_t mirror_cap;
  step_accept_block(c, b, &mirror_rec, &mirror_cnt, &mirror_cap);
  const int sel_before = c->accept_sel;
  c->accept_sel = 2;
  c->accept_armed = false;
  c->accept_streamed = false;
  rc = sf_spec_launch(c, fo.cand, fo.count);
  c->accept_sel = sel_before;
  b.armed = c->accept_armed;
  b.streamed = c->accept_streamed;
  b.pairs = (int32_t)c->spec.grid;
  if (rc != SF_OK) return rc;
  // the second stream: exact distances -> row minima (with each minimum's candidate index) -> argsort + walk
  {
    SF_HIP(c, hipStreamWaitEvent(c->aux, c->ev_filter, 0));
    const hipStream_t lane_stream = c->stream;
    c->stream = c->aux;                      // (the launchers queue on, and bracket for, the handle's current stream)
    rc = sf_nn_minima_of_candidates_dev(c, fo, d_min, d_arg, d_status, d_cand, d_arg64, c->store.kcap >= 256);
    if (rc == SF_OK)
      rc = sf_nn_walk_dev(c, d_min, d_arg, d_status, n_l, n_r, c->params.netvlad_distance, c->params.netvlad_max_matches_nb,
                          lim, nullptr, nullptr, b.walk_matches, b.walk_n, b.walk_status, d_cand, b.walk_slots, fo.count,
                          c->spec.grid);
    c->stream = lane_stream;
    if (rc != SF_OK) { (void)hipStreamSynchronize(c->aux); return rc; }
    SF_HIP(c, hipEventRecord(c->ev_walk, c->aux));
  }
  SF_HIP(c, hipStreamWaitEvent(c->stream, c->ev_walk, 0));      // the step is done when both streams are
  if (!b.streamed)      // (step_issue_device picks this form only where the launch streams: a plan / arm mismatch)
    return sf_fail(c, SF_EHIP, "speculative step: the verification launch did not arm the accepted-result stream");
  SF_HIP(c, hipEventRecord(b.done, c->stream));
  return SF_OK;
}

// Everything on one stream: NN kernels -> row minima -> argsort + walk -> verification of the walk's matches, taken from
// the device list.  Any query shape (the reference's cadence of 20 matches per tick included), both NN precisions.
static int step_issue_serial(sf_context* c, sf_context::StepBlock& b, int32_t slot_base_other, int32_t slot_base_local, int lim) {
  const int n_l = c->nn_local.n, n_r = c->nn_recv.n;
  int rc;
  const size_t min_bytes = ((size_t)n_l * 8 + 63) & ~(size_t)63, arg_bytes = ((size_t)n_l * 4 + 63) & ~(size_t)63;
  if ((rc = sf_buf_reserve(c, c->step_nn, min_bytes + arg_bytes + 64)) != SF_OK) return rc;
  double* d_min = (double*)c->step_nn.p;
  int32_t* d_arg = (int32_t*)((char*)c->step_nn.p + min_bytes);
  int32_t* d_status = (int32_t*)((char*)c->step_nn.p + min_bytes + arg_bytes);
  unsigned* d_count = (unsigned*)b.dev.p;                       // {matches, -, -, -, accept slot counter, ...}
  void* d_match_rc = (char*)b.dev.p + 64;
  if ((rc = sf_nn_row_minima_dev(c, d_min, d_arg, d_status)) != SF_OK) return rc;
  if ((rc = sf_nn_walk_dev(c, d_min, d_arg, d_status, n_l, n_r, c->params.netvlad_distance, c->params.netvlad_max_matches_nb,
                           lim, d_match_rc, d_count, b.walk_matches, b.walk_n, b.walk_status)) != SF_OK) return rc;
  // verification of the walk's matches: `lim` pair slots, those past the device-side count are void
  c->spec.grid = (unsigned)lim;
  c->spec.slot_other = slot_base_other;
  c->spec.slot_local = slot_base_local;
  if ((rc = sf_buf_reserve(c, c->spec_from, (size_t)lim * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->spec_to, (size_t)lim * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->spec_results, (size_t)lim * sizeof(sf_result))) != SF_OK) return rc;
  sf_result* mirror_rec; uint32_t* mirror_cnt; int32_t mirror_cap;
  step_accept_block(c, b, &mirror_rec, &mirror_cnt, &mirror_cap);
  const int sel_before = c->accept_sel;
  c->accept_sel = 2;
  c->accept_armed = false;
  c->accept_streamed = false;
  rc = sf_spec_launch(c, d_match_rc, d_count);
  c->accept_sel = sel_before;
  b.armed = c->accept_armed;
  b.streamed = c->accept_streamed;
  b.pairs = lim;
  if (rc != SF_OK) return rc;
  if (!b.streamed) {
    // a launch shape the stream does not cover (stage kernels, more than one chunk, a mirror smaller than the query):
    // ordered compaction of the `lim` slots, match order -- the void slots past the match count carry success = 0
    if (lim > b.cap) return sf_fail(c, SF_ERANGE, "sf_step_issue: %d pair slots exceed the block's %d records", lim, b.cap);
    if ((rc = compact_launch(c, (const sf_result*)c->spec_results.p, lim, b.records, b.flags, b.count, nullptr, mirror_rec,
                             nullptr, (int32_t*)mirror_cnt, mirror_cap)) != SF_OK) return rc;
  }
  SF_HIP(c, hipEventRecord(b.done, c->stream));
  return SF_OK;
}

static int step_issue_device(sf_context* c, sf_context::StepBlock& b, int32_t slot_base_other, int32_t slot_base_local) {
  const int n_l = c->nn_local.n;
  const int lim = std::min(n_l, c->params.netvlad_max_matches_nb);     // the walk looks at `lim` rows: at most `lim` matches
  b.device_walk = true;
  b.speculative = false;
  b.streamed = false; b.armed = false; b.n = 0; b.pairs = 0;
  *b.walk_status = 0;
  *b.walk_n = 0;
  if (lim <= 0) {                                   // (netvlad_max_matches_nb = 0: the walk returns nothing)
    SF_HIP(c, hipEventRecord(b.done, c->stream));
    return SF_OK;
  }
  // the speculative form where round 3 speculated (the walk may return every local row, filter path) and where every
  // verified slot can stream: one chunk, a chain-type launch, a block / mirror with a record slot for every candidate slot
  if (c->step_speculate && c->params.nn_precision == 1 && lim >= n_l) {
    const int grid = n_l + n_l / 8 + 256;
    const VerifyPlan plan = verify_plan(c, sf_store_view(c->store), grid);
    sf_result* const mirror_rec = c->step_mirror_records[b.parity];
    const int32_t cap = std::min(b.cap, mirror_rec ? c->step_mirror_cap : b.cap);
    if (plan.streams() && cap >= grid) {
      b.speculative = true;
      return step_issue_speculative(c, b, slot_base_other, slot_base_local, lim);
    }
  }
  return step_issue_serial(c, b, slot_base_other, slot_base_local, lim);
}

extern "C" int sf_step_issue(sf_handle c, int32_t slot_base_other, int32_t slot_base_local) {
  if (!c) return SF_EINVAL;
  if (c->step_inflight >= c->step_depth)
    return sf_fail(c, SF_EINVAL, "%d steps are in flight (SF_OPT_STEP_DEPTH): call sf_step_retire first", c->step_inflight);
  if (c->nn_local.n <= 0 || c->nn_recv.n <= 0)
    return sf_fail(c, SF_EINVAL, "empty descriptor database (data_handler.py:308 guards this case)");
  SF_HIP(c, hipSetDevice(c->device));
  const bool mirrored = c->step_mirror_records[0] != nullptr;
  if (mirrored) {
    // A mirror has ONE caller buffer (and one caller-zeroed counter) per parity: step k + 2 writes where step k wrote.  The
    // ring's depth (default 6) would let step k + 2 zero and overwrite the mirror before step k is retired and its collective
    // enqueued, so with a mirror set no more steps may be in flight than the mirror has buffers.
    const int mirror_buffers = (c->step_mirror_records[1] && c->step_mirror_records[1] != c->step_mirror_records[0]) ? 2 : 1;
    if (c->step_inflight >= mirror_buffers)
      return sf_fail(c, SF_EINVAL, "%d step(s) in flight with a %d-buffer mirror set (sf_step_mirror%s): retire before issuing -- "
                                   "step k + %d would overwrite the records of step k", c->step_inflight, mirror_buffers,
                     mirror_buffers == 2 ? "_pair" : "", mirror_buffers);
  }
  int lanes = (c->step_overlap && !c->overlap && (!mirrored || c->step_mirror_lanes)) ? c->step_lanes : 1;
  if (mirrored) lanes = std::min(lanes, 2);          // (a mirror's buffer and its collective live on the stream of its parity)
  if (c->params.nn_precision == 0) lanes = 1;        // (the fp32-ranking path keeps its partial minima in ONE workspace)
  const int lane = (int)(c->step_seq % (uint64_t)lanes);
  sf_context::StepBlock& b = c->step_blocks[c->step_seq % (uint64_t)step_ring(c)];
  const int n_l = c->nn_local.n;
  int rc;
  // every slot of a speculative verification may be accepted: the block holds them all (see arm_accept_stream)
  if ((rc = step_block_reserve(c, b, n_l + n_l / 8 + 256)) != SF_OK) return rc;
  b.parity = (int)(c->step_seq & 1);
  b.slot_other = slot_base_other; b.slot_local = slot_base_local;
  b.settled = false; b.settle_rc = SF_OK;
  c->in_overlapped_step = lanes > 1;            // (sf_use_split: the form the verification takes)
  const bool device = c->step_device_walk && !c->overlap && c->store.slots > 0;
  c->cur_lane = lane;
  if (lane > 0 && (rc = lane_enter(c, lane)) != SF_OK) { c->in_overlapped_step = false; return rc; }
  if (b.copy_pending) {            // (sf_memcpy_device_async out of this block's records, possibly on a stream of the caller's)
    hipError_t e = hipStreamWaitEvent(c->stream, b.copied, 0);
    if (e != hipSuccess) rc = sf_fail(c, SF_EHIP, "hipStreamWaitEvent(copy of d_records) -> %s", hipGetErrorString(e));
    b.copy_pending = false;
  }
  // state every lane reads (fp16 copies, coefficients, masks) prepared by another lane since this one last looked?
  if (c->lane_seen_prep[lane] != c->prep_epoch && c->ev_prep) {
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_prep, 0);
    if (e != hipSuccess) rc = sf_fail(c, SF_EHIP, "hipStreamWaitEvent -> %s", hipGetErrorString(e));
  }
  c->lane_seen_prep[lane] = c->prep_epoch;
  const uint64_t prep_before = c->prep_count;
  if (rc == SF_OK) rc = device ? step_issue_device(c, b, slot_base_other, slot_base_local)
                               : step_issue_sync(c, b, slot_base_other, slot_base_local);
  if (c->prep_count != prep_before) {
    if (!c->ev_prep) (void)hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming);
    if (c->ev_prep) (void)hipEventRecord(c->ev_prep, c->stream);
    c->prep_epoch += 1;
    c->lane_seen_prep[lane] = c->prep_epoch;
  }
  if (rc != SF_OK) {
    // nothing of a failed issue may stay behind: whatever was queued is waited for and the block's streamed entries are
    // reset, so the next step starts from a clean block
    (void)hipStreamSynchronize(c->stream);
    for (int32_t r = 0; r < b.cap && b.index[r] >= 0; ++r) b.index[r] = -1;
  }
  if (lane > 0) lane_swap(c, lane);
  c->in_overlapped_step = false;
  if (rc != SF_OK) return rc;
  b.issued = true;
  c->step_seq += 1;
  c->step_inflight += 1;
  return SF_OK;
}

// Waits for a step and, if its device walk reported a candidate set too dense for the filter level, runs the query again
// on the synchronous path (the handle is idle by then: every step in flight is waited for first, since the ladder rewrites
// state all lanes read).  Idempotent.
static int step_settle(sf_context* c, sf_context::StepBlock& b) {
  if (b.settled) return b.settle_rc;
  b.settled = true;
  hipError_t e = hipEventSynchronize(b.done);
  if (e != hipSuccess) return b.settle_rc = sf_fail(c, SF_EHIP, "hipEventSynchronize(step) -> %s", hipGetErrorString(e));
  if (!b.device_walk || *b.walk_status == 0) return b.settle_rc = SF_OK;
  for (auto& o : c->step_blocks)
    if (o.issued && o.done) (void)hipEventSynchronize(o.done);
  // (what the void verification may have streamed: nothing -- the walk emitted no match -- but the entries are the next
  //  query's, so make sure)
  for (int32_t r = 0; r < b.cap && b.index[r] >= 0; ++r) b.index[r] = -1;
  const bool mirrored = c->step_mirror_records[b.parity] != nullptr;
  c->in_overlapped_step = false;
  int rc = step_issue_sync(c, b, b.slot_other, b.slot_local);        // (on the handle's own stream and buffers)
  if (rc == SF_OK && (e = hipEventSynchronize(b.done)) != hipSuccess)
    rc = sf_fail(c, SF_EHIP, "hipEventSynchronize(step fallback) -> %s", hipGetErrorString(e));
  if (rc == SF_OK && mirrored)
    rc = sf_fail(c, SF_ERANGE, "sf_step_retire: the NN candidate set outgrew the filter level while a mirror was set -- the "
                               "mirror (parity %d) missed this step's records.  The level is settled now; the step counter has "
                               "advanced, so a re-issued step writes the mirror of parity %d: retire everything in flight, "
                               "re-zero both counters and issue the step again", b.parity, (int)(c->step_seq & 1));
  return b.settle_rc = rc;
}

static int step_settle_all(sf_context* c) {
  int rc = SF_OK;
  for (int k = c->step_inflight; k >= 1; --k) {          // oldest first
    sf_context::StepBlock& b = c->step_blocks[(c->step_seq - (uint64_t)k) % (uint64_t)step_ring(c)];
    const int r = step_settle(c, b);
    if (rc == SF_OK) rc = r;
  }
  return rc;
}

extern "C" int sf_step_retire(sf_handle c, sf_step_result* out) {
  if (!c || !out) return SF_EINVAL;
  memset(out, 0, sizeof(*out));
  if (c->step_inflight <= 0) return sf_fail(c, SF_EINVAL, "sf_step_retire: no step in flight");
  sf_context::StepBlock& b = c->step_blocks[(c->step_seq - (uint64_t)c->step_inflight) % (uint64_t)step_ring(c)];   // the OLDEST
  SF_HIP(c, hipSetDevice(c->device));
  int rc = step_settle(c, b);                    // its verification (and compaction) has left the device
  // whatever happens below, the step leaves the pipeline and its block is clean for its next use
  b.issued = false;
  c->step_inflight -= 1;
  int32_t n_streamed = 0;
  if (b.armed) {
    // streamed records: completion order, one per ACCEPTED verified slot; the used entries of the index list are its
    // prefix.  They are reset here for the block's next step -- also when the query fell back and never read them.
    while (n_streamed < b.cap && b.index[n_streamed] >= 0) ++n_streamed;
    if (b.streamed) b.rec_of_slot.assign((size_t)std::max(b.pairs, 1), -1);
    for (int32_t r = 0; r < n_streamed; ++r) {
      const int32_t slot = b.index[r];
      if (b.streamed && slot < b.pairs) b.rec_of_slot[slot] = r;
      b.index[r] = -1;
    }
  }
  if (rc != SF_OK) return rc;
  const sf_match* matches = b.matches.data();
  int n = b.n;
  if (b.device_walk) {
    n = *b.walk_n;
    if (n < 0 || n > std::max(b.pairs, c->nn_local.n)) return sf_fail(c, SF_EHIP, "sf_step_retire: the device walk reports %d matches of %d slots", n, b.pairs);
    matches = b.walk_matches;
  }
  b.record_of_match.assign((size_t)std::max(n, 1), -1);
  int32_t n_records = 0, n_accepted = 0;
  if (b.streamed) {
    n_records = n_streamed;
    for (int i = 0; i < n; ++i) {
      // (serial device step: pair slot i IS match i; speculative: the slot of the match's candidate)
      const int32_t slot = b.device_walk ? (b.speculative ? b.walk_slots[i] : i) : b.slot_of_match[i];
      const int32_t r = (slot >= 0 && slot < b.pairs) ? b.rec_of_slot[slot] : -1;
      b.record_of_match[i] = r;
      n_accepted += r >= 0;
    }
  } else if (n > 0) {
    n_records = *b.count;
    if (n_records < 0) return sf_fail(c, SF_EHIP, "sf_step_retire: the ordered compaction's look-back timed out");
    int32_t run = 0;
    for (int i = 0; i < n; ++i) b.record_of_match[i] = b.flags[i] ? run++ : -1;
    n_accepted = run;
    if (run != n_records) return sf_fail(c, SF_EHIP, "sf_step_retire: %d flags set, %d records compacted", run, n_records);
  }
  out->matches = matches;
  out->n_matches = n;
  out->record_of_match = b.record_of_match.data();
  out->records = b.records;
  out->d_records = c->step_mirror_records[b.parity] ? nullptr : (const sf_result*)b.dev_records.p;
  out->n_records = n_records;
  out->n_accepted = n_accepted;
  out->streamed = b.streamed ? 1 : 0;
  return SF_OK;
}

extern "C" int sf_memcpy_device_async(sf_handle c, void* d_dst, const void* d_src, size_t bytes, void* hip_stream) {
  if (!c || (bytes > 0 && (!d_dst || !d_src))) return SF_EINVAL;
  if (bytes == 0) return SF_OK;
  SF_HIP(c, hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  SF_HIP(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, s));
  // out of a step block's record buffer: the block's next step must not overwrite it before this copy has run
  for (auto& b : c->step_blocks) {
    const char* lo = (const char*)b.dev_records.p;
    if (!lo || (const char*)d_src < lo || (const char*)d_src >= lo + b.dev_records.bytes) continue;
    if (!b.copied) SF_HIP(c, hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
    SF_HIP(c, hipEventRecord(b.copied, s));
    b.copy_pending = true;
    break;
  }
  return SF_OK;
}

extern "C" int sf_last_match_results(sf_handle c, const sf_result** d_results, const int32_t** index, int32_t* n) {
  if (!c || !d_results || !index || !n) return SF_EINVAL;
  *d_results = c->last_results; *index = c->last_results_index; *n = c->last_results_n;
  return SF_OK;
}

extern "C" int sf_compact_accepted_indexed_device_async(sf_handle c, const sf_result* d_results, const int32_t* index,
                                                        int32_t n, sf_result* d_accepted, uint8_t* d_flags,
                                                        int32_t* d_n_accepted) {
  if (!c || n < 0 || !d_n_accepted || (n > 0 && (!d_results || !d_accepted))) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  if (n == 0) {
    SF_HIP(c, hipMemsetAsync(d_n_accepted, 0, 4, c->stream));
    return SF_OK;
  }
  return compact_launch(c, d_results, n, d_accepted, d_flags, d_n_accepted, index);
}

extern "C" int sf_compact_accepted_indexed_mirrored_device_async(sf_handle c, const sf_result* d_results,
                                                                 const int32_t* index, int32_t n, sf_result* d_accepted,
                                                                 uint8_t* d_flags, int32_t* d_n_accepted,
                                                                 sf_result* d_accepted2, uint8_t* d_flags2,
                                                                 int32_t* d_n_accepted2) {
  if (!c || n < 0 || !d_n_accepted || !d_n_accepted2 || (n > 0 && (!d_results || !d_accepted || !d_accepted2))) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  if (n == 0) {
    SF_HIP(c, hipMemsetAsync(d_n_accepted, 0, 4, c->stream));
    SF_HIP(c, hipMemsetAsync(d_n_accepted2, 0, 4, c->stream));
    return SF_OK;
  }
  return compact_launch(c, d_results, n, d_accepted, d_flags, d_n_accepted, index, d_accepted2, d_flags2, d_n_accepted2);
}

extern "C" int sf_compact_accepted_device(sf_handle c, const sf_result* d_results, int32_t n, sf_result* d_accepted,
                                          uint8_t* d_flags, int32_t* n_accepted) {
  if (!c || n < 0 || !n_accepted || (n > 0 && (!d_results || !d_accepted))) return SF_EINVAL;
  *n_accepted = 0;
  if (n == 0) return SF_OK;
  SF_HIP(c, hipSetDevice(c->device));
  const int chunks = (n + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
  int rc;
  if ((rc = sf_buf_reserve(c, c->compact_scratch, (size_t)(chunks + 2) * 8)) != SF_OK) return rc;
  int32_t* d_count = (int32_t*)((char*)c->compact_scratch.p + (size_t)(chunks + 1) * 8);   // behind the chunk states
  if ((rc = compact_launch(c, d_results, n, d_accepted, d_flags, d_count)) != SF_OK) return rc;
  if (!c->count_pinned && hipHostMalloc((void**)&c->count_pinned, 64, hipHostMallocDefault) != hipSuccess)
    return sf_fail(c, SF_ENOMEM, "hipHostMalloc(64) failed");
  SF_HIP(c, hipMemcpyAsync(c->count_pinned, d_count, 4, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (*c->count_pinned < 0) return sf_fail(c, SF_EHIP, "ordered compaction: the look-back over earlier chunks timed out");
  *n_accepted = *c->count_pinned;
  return SF_OK;
}

static int verify_host_indices(sf_context* c, const Store& st, const int32_t* from, const int32_t* to, int n,
                               sf_result* out) {
  if (n == 0) return SF_OK;
  for (int i = 0; i < n; ++i)
    if (from[i] < 0 || from[i] >= st.slots || to[i] < 0 || to[i] >= st.slots)
      return sf_fail(c, SF_ERANGE, "pair %d: slot (%d,%d) outside the store (%d slots)", i, from[i], to[i], st.slots);
  int rc;
  if ((rc = sf_buf_reserve(c, c->pair_from, (size_t)n * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->pair_to, (size_t)n * 4)) != SF_OK) return rc;
  if ((rc = sf_buf_reserve(c, c->results, (size_t)n * sizeof(sf_result))) != SF_OK) return rc;
  SF_HIP(c, hipMemcpyAsync(c->pair_from.p, from, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  SF_HIP(c, hipMemcpyAsync(c->pair_to.p, to, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if ((rc = verify_device(c, st, (const int32_t*)c->pair_from.p, (const int32_t*)c->pair_to.p, n,
                          (sf_result*)c->results.p)) != SF_OK) return rc;
  SF_HIP(c, hipMemcpyAsync(out, c->results.p, (size_t)n * sizeof(sf_result), hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  return SF_OK;
}

extern "C" int sf_verify_pairs(sf_handle c, const int32_t* from_slot, const int32_t* to_slot, int32_t n,
                               sf_result* out) {
  if (!c || n < 0 || (n > 0 && (!from_slot || !to_slot || !out))) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return verify_host_indices(c, c->store, from_slot, to_slot, n, out);
}

extern "C" int sf_estimate_transform_batch(sf_handle c, const sf_features* from, const sf_features* to,
                                           int32_t n, sf_result* out) {
  if (!c || n < 0 || (n > 0 && (!from || !to || !out))) return SF_EINVAL;
  if (n == 0) return SF_OK;
  SF_HIP(c, hipSetDevice(c->device));
  int rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = validate_features(c, from + i)) != SF_OK) return rc;
    if ((rc = validate_features(c, to + i)) != SF_OK) return rc;
    if (from[i].rows > 0 && to[i].rows > 0 && from[i].cols != to[i].cols)
      return sf_fail(c, SF_EINVAL, "pair %d: descriptor widths differ (%d vs %d; myRegistrationVis.cpp:683)", i,
                     (int)from[i].cols, (int)to[i].cols);
  }
  SF_HIP(c, hipStreamSynchronize(c->stream));
  c->scratch.slots = 0;
  std::vector<int32_t> fi(n), ti(n);
  std::vector<const sf_features*> all(2 * (size_t)n);
  for (int i = 0; i < n; ++i) { all[2 * i] = from + i; all[2 * i + 1] = to + i; }
  int first = 0;
  if ((rc = store_add_host_batch(c, c->scratch, all.data(), 2 * n, &first)) != SF_OK) return rc;
  for (int i = 0; i < n; ++i) { fi[i] = first + 2 * i; ti[i] = first + 2 * i + 1; }
  return verify_host_indices(c, c->scratch, fi.data(), ti.data(), n, out);
}

extern "C" int sf_estimate_transform(sf_handle c, const sf_features* from, const sf_features* to, sf_result* out) {
  return sf_estimate_transform_batch(c, from, to, 1, out);
}

extern "C" int sf_debug_correspondences(sf_handle c, int32_t pair, int32_t pass, uint16_t* from_idx,
                                        uint16_t* to_idx, int32_t cap, int32_t* n_out) {
  if (!c || !n_out || pair < 0 || (pass != 1 && pass != 2)) return SF_EINVAL;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (c->ws_split > 0 && pair >= c->ws_split) {   // second half of a two-stream batch: the shadow workspace
    pair -= c->ws_split;
    c = c->twin;
  }
  if (pair >= c->ws_pairs) return SF_EINVAL;
  if (!c->last_lists_valid)
    return sf_fail(c, SF_EINVAL, "the fused kernel keeps correspondence lists in LDS: set SF_OPT_DEBUG_CORR (or "
                                 "SF_DEBUG_CORR=1) before the verification call");
  CorrHeader h;
  const Buf& hb = pass == 1 ? c->hdr1 : c->hdr2;
  const Buf& cb = pass == 1 ? c->corr1 : c->corr2;
  SF_HIP(c, hipMemcpy(&h, (const CorrHeader*)hb.p + pair, sizeof(h), hipMemcpyDeviceToHost));
  int n = std::min(h.n_corr, cap);
  std::vector<uint32_t> tmp(std::max(n, 1));
  if (n > 0) SF_HIP(c, hipMemcpy(tmp.data(), (const uint32_t*)cb.p + (size_t)pair * c->ws_kcap, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    if (from_idx) from_idx[i] = (uint16_t)(tmp[i] & 0xFFFFu);
    if (to_idx) to_idx[i] = (uint16_t)(tmp[i] >> 16);
  }
  *n_out = h.n_corr;
  return SF_OK;
}

// Pass state of pair `pair` of the LAST verification (diagnostics; needs SF_OPT_DEBUG_CORR like the lists): the pass's
// pose (row-major 3 x 4, p_from = T p_to, all zero when null), is_null / inliers / matches.
extern "C" int sf_debug_pass_state(sf_handle c, int32_t pair, int32_t pass, float* T12, int32_t* is_null, int32_t* inliers,
                                   int32_t* matches) {
  if (!c || pair < 0 || (pass != 1 && pass != 2)) return SF_EINVAL;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if (c->ws_split > 0 && pair >= c->ws_split) { pair -= c->ws_split; c = c->twin; }
  if (pair >= c->ws_pairs) return SF_EINVAL;
  if (!c->last_lists_valid) return sf_fail(c, SF_EINVAL, "set SF_OPT_DEBUG_CORR before the verification call");
  PassState ps;
  SF_HIP(c, hipMemcpy(&ps, (const PassState*)(pass == 1 ? c->pass1.p : c->pass2.p) + pair, sizeof(ps), hipMemcpyDeviceToHost));
  if (T12) memcpy(T12, ps.T, sizeof(ps.T));
  if (is_null) *is_null = ps.is_null;
  if (inliers) *inliers = ps.inliers;
  if (matches) *matches = ps.matches;
  return SF_OK;
}

extern "C" int sf_debug_counters(sf_handle c, unsigned long long* out, int32_t n) {
  if (!c || !out || n < 0) return SF_EINVAL;
  memset(out, 0, (size_t)n * 8);
  if (!c->trace.p) return SF_OK;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  if ((size_t)n * 8 > c->trace.bytes) return SF_ERANGE;
  SF_HIP(c, hipMemcpy(out, c->trace.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return SF_OK;
}
// (experiment) per-point records of the guided pass of pair `pair` of the last verification: two planes of kcap words
extern "C" int sf_debug_guided_points(sf_handle c, int32_t pair, unsigned long long* plane0, unsigned long long* plane1,
                                      int32_t* kcap_out) {
  if (!c || pair < 0 || pair >= c->ws_pairs || !c->trace.p) return SF_EINVAL;
  SF_HIP(c, hipStreamSynchronize(c->stream));
  const size_t k = (size_t)c->ws_kcap, np = (size_t)c->ws_pairs;
  if (c->trace.bytes < 4096 + 2 * np * k * 8) return SF_EINVAL;
  const char* base = (const char*)c->trace.p + 4096;
  SF_HIP(c, hipMemcpy(plane0, base + ((size_t)pair * k) * 8, k * 8, hipMemcpyDeviceToHost));
  SF_HIP(c, hipMemcpy(plane1, base + ((np + (size_t)pair) * k) * 8, k * 8, hipMemcpyDeviceToHost));
  if (kcap_out) *kcap_out = (int32_t)k;
  return SF_OK;
}

// ---- separator records ----------------------------------------------------------------------------
extern "C" int sf_pack_separators(const sf_result* res, int32_t n, int8_t robot_from, int8_t robot_to,
                                  const int16_t* kf_from, const int16_t* kf_to, const int16_t* frame_from,
                                  const int16_t* frame_to, sf_separator* out) {
  if (n < 0 || (n > 0 && (!res || !out))) return SF_EINVAL;
  for (int i = 0; i < n; ++i) {
    sf_separator s;
    memset(&s, 0, sizeof(s));
    s.robot_from_id = robot_from;
    s.robot_to_id = robot_to;
    s.kf_id_from = kf_from ? kf_from[i] : 0;
    s.kf_id_to = kf_to ? kf_to[i] : 0;
    s.frame_id_from = frame_from ? frame_from[i] : 0;
    s.frame_id_to = frame_to ? frame_to[i] : 0;
    s.transform_est_success = res[i].success;
    memcpy(s.position, res[i].position, sizeof(s.position));
    memcpy(s.orientation, res[i].orientation, sizeof(s.orientation));
    memcpy(s.covariance, res[i].covariance, sizeof(s.covariance));
    out[i] = s;
  }
  return SF_OK;
}

// ---- measurement ----------------------------------------------------------------------------------
// (the launches of the second stream of a two-stream batch are booked on the shadow context and reported
//  together with the handle's own)
extern "C" int sf_prof_enable(sf_handle c, int on) {
  if (!c) return SF_EINVAL;
  prof_resolve(c);
  c->prof = on != 0;
  if (c->twin) { prof_resolve(c->twin); c->twin->prof = c->prof; }
  return SF_OK;
}

extern "C" int sf_prof_select(sf_handle c, uint32_t kernel_mask) {
  if (!c) return SF_EINVAL;
  prof_resolve(c);
  c->prof_mask = kernel_mask;
  if (c->twin) { prof_resolve(c->twin); c->twin->prof_mask = kernel_mask; }
  return SF_OK;
}

extern "C" int sf_prof_reset(sf_handle c) {
  if (!c) return SF_EINVAL;
  prof_resolve(c);
  for (auto& s : c->prof_slots) s = ProfSlot();
  if (c->twin) {
    prof_resolve(c->twin);
    for (auto& s : c->twin->prof_slots) s = ProfSlot();
  }
  return SF_OK;
}

extern "C" int sf_prof_get(sf_handle c, int kernel, int64_t* launches, double* total_ms) {
  if (!c || kernel < 0 || kernel >= SF_K_COUNT) return SF_EINVAL;
  prof_resolve(c);
  int64_t n = c->prof_slots[kernel].launches;
  double ms = c->prof_slots[kernel].total_ms;
  if (c->twin) {
    prof_resolve(c->twin);
    n += c->twin->prof_slots[kernel].launches;
    ms += c->twin->prof_slots[kernel].total_ms;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  return SF_OK;
}

extern "C" int sf_set_option(sf_handle c, int32_t option, int32_t value) {
  if (!c) return SF_EINVAL;
  switch (option) {
    case SF_OPT_MATCH_MFMA: c->match_mfma = value != 0; return SF_OK;
    case SF_OPT_FUSED: c->fused = value != 0; return SF_OK;
    case SF_OPT_OVERLAP: c->overlap = value != 0; return SF_OK;
    case SF_OPT_CHAIN_WAVES: return SF_OK;   // (round 1's narrower chains are gone: accepted, no effect)
    case SF_OPT_DEBUG_CORR: c->debug_corr = value != 0; return SF_OK;
    case SF_OPT_NN_FULL_FILTER: c->nn_force_full = value != 0; c->nn_coef_level = -1; return SF_OK;
    case SF_OPT_STEP_SPLIT: c->split_auto = value != 0; return SF_OK;
    case SF_OPT_STEP_OVERLAP:
      if (c->step_inflight) return sf_fail(c, SF_EINVAL, "SF_OPT_STEP_OVERLAP cannot change while steps are in flight");
      c->step_overlap = value != 0;
      return SF_OK;
    case SF_OPT_STEP_DEPTH:
      if (c->step_inflight) return sf_fail(c, SF_EINVAL, "SF_OPT_STEP_DEPTH cannot change while steps are in flight");
      if (value < 1 || value > SF_STEP_MAX_DEPTH) return sf_fail(c, SF_ERANGE, "SF_OPT_STEP_DEPTH %d not in 1..%d", value, SF_STEP_MAX_DEPTH);
      c->step_depth = value;
      c->step_seq = 0;
      return SF_OK;
    case SF_OPT_STEP_LANES:
      if (c->step_inflight) return sf_fail(c, SF_EINVAL, "SF_OPT_STEP_LANES cannot change while steps are in flight");
      if (value < 1 || value > SF_STEP_MAX_LANES) return sf_fail(c, SF_ERANGE, "SF_OPT_STEP_LANES %d not in 1..%d", value, SF_STEP_MAX_LANES);
      c->step_lanes = value;
      return SF_OK;
    case SF_OPT_STEP_SPECULATE:
      if (c->step_inflight) return sf_fail(c, SF_EINVAL, "SF_OPT_STEP_SPECULATE cannot change while steps are in flight");
      c->step_speculate = value != 0;
      return SF_OK;
    case SF_OPT_STEP_DEVICE_WALK:
      if (c->step_inflight) return sf_fail(c, SF_EINVAL, "SF_OPT_STEP_DEVICE_WALK cannot change while steps are in flight");
      c->step_device_walk = value != 0;
      return SF_OK;
    default: return sf_fail(c, SF_EINVAL, "unknown option %d", option);
  }
}

// ---- NN stage entry points (implementation in k_nn.hip) --------------------------------------------
extern "C" int sf_nn_append_local(sf_handle c, const double* desc, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_local, desc, n, dim, 0);
}
extern "C" int sf_nn_append_received(sf_handle c, const double* desc, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_recv, desc, n, dim, 0);
}
extern "C" int sf_nn_append_local_f16_device(sf_handle c, const uint16_t* d, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_local, d, n, dim, 2);
}
extern "C" int sf_nn_append_received_f16_device(sf_handle c, const uint16_t* d, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_recv, d, n, dim, 2);
}
extern "C" int sf_nn_append_local_f32_device(sf_handle c, const float* d, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_local, d, n, dim, 1);
}
extern "C" int sf_nn_append_received_f32_device(sf_handle c, const float* d, int32_t n, int32_t dim) {
  if (!c) return SF_EINVAL;
  return sf_nn_append(c, c->nn_recv, d, n, dim, 1);
}

extern "C" int sf_nn_sizes(sf_handle c, int32_t* n_local, int32_t* n_received) {
  if (!c) return SF_EINVAL;
  if (n_local) *n_local = c->nn_local.n;
  if (n_received) *n_received = c->nn_recv.n;
  return SF_OK;
}

extern "C" int sf_nn_mark_local_used(sf_handle c, int32_t idx) {
  if (!c) return SF_EINVAL;
  if (idx < 0 || idx >= c->nn_local.n) return sf_fail(c, SF_ERANGE, "local index %d outside [0,%d)", idx, c->nn_local.n);
  // steps in flight were issued on the masks as they are: they are waited for first (arguments validated before the drain;
  // a settle error -- a step whose re-run failed -- is this call's error: the mask is not touched then)
  if (c->step_inflight) { const int rc = sf_lanes_touch(c, false); if (rc != SF_OK) return rc; }
  if ((int)c->mask_local.size() < c->nn_local.n) c->mask_local.resize(c->nn_local.n, 0);
  c->mask_local[idx] = 1;
  c->masks_dirty = true;
  return SF_OK;
}

extern "C" int sf_nn_mark_other_used(sf_handle c, int32_t idx) {
  if (!c) return SF_EINVAL;
  if (idx < 0 || idx >= c->nn_recv.n) return sf_fail(c, SF_ERANGE, "other index %d outside [0,%d)", idx, c->nn_recv.n);
  if (c->step_inflight) { const int rc = sf_lanes_touch(c, false); if (rc != SF_OK) return rc; }
  if ((int)c->mask_other.size() < c->nn_recv.n) c->mask_other.resize(c->nn_recv.n, 0);
  c->mask_other[idx] = 1;
  c->masks_dirty = true;
  return SF_OK;
}

extern "C" int sf_nn_ignore_pair(sf_handle c, int32_t il, int32_t io) {
  if (!c) return SF_EINVAL;
  if (il < 0 || il >= c->nn_local.n || io < 0 || io >= c->nn_recv.n)
    return sf_fail(c, SF_ERANGE, "pair (%d,%d) outside the %d x %d distance matrix", il, io, c->nn_local.n, c->nn_recv.n);
  if (c->step_inflight) { const int rc = sf_lanes_touch(c, false); if (rc != SF_OK) return rc; }
  c->ignored.push_back(il);
  c->ignored.push_back(io);
  c->masks_dirty = true;
  return SF_OK;
}

extern "C" int sf_nn_reset(sf_handle c) {
  if (!c) return SF_EINVAL;
  (void)sf_lanes_touch(c, true);
  SF_HIP(c, hipStreamSynchronize(c->stream));
  // the row buffers are sized, pitched and zero-padded for the old dimension: release them, the next append
  // re-allocates for its own (nn_reserve); the fp16 copies and cached filter coefficients go with them
  for (NNDb* db : {&c->nn_local, &c->nn_recv}) {
    buf_free(db->rows); buf_free(db->norms); buf_free(db->rows_h); buf_free(db->norms_k);
    db->n = 0; db->cap = 0; db->ld = 0; db->h_n = -1; db->h_ld = 0; db->h_kprefix = 0;
  }
  c->nn_coef_level = -1;
  c->nn_level = 0;
  c->nn_level_cooldown = 32;
  c->nn_dim = 0;
  c->mask_local.clear();
  c->mask_other.clear();
  c->ignored.clear();
  c->masks_dirty = true;
  return SF_OK;
}

extern "C" int sf_nn_set_precision(sf_handle c, int32_t nn_precision) {
  if (!c) return SF_EINVAL;
  if (nn_precision != 0 && nn_precision != 1) return sf_fail(c, SF_EINVAL, "nn_precision must be 0 or 1");
  c->params.nn_precision = nn_precision;
  return SF_OK;
}

extern "C" int sf_nn_find_matches(sf_handle c, sf_match* out, int32_t cap, int32_t* n_out) {
  if (!c || !n_out || cap < 0 || (cap > 0 && !out)) return SF_EINVAL;
  *n_out = 0;
  if (c->nn_local.n <= 0 || c->nn_recv.n <= 0)
    return sf_fail(c, SF_EINVAL, "empty descriptor database (data_handler.py:308 guards this case)");
  SF_HIP(c, hipSetDevice(c->device));
  return sf_nn_run(c, out, cap, n_out);
}

extern "C" int sf_nn_row_minima_device(sf_handle c, double* d_row_min, int32_t* d_row_arg, int32_t* d_status) {
  if (!c || !d_row_min || !d_row_arg || !d_status) return SF_EINVAL;
  SF_HIP(c, hipSetDevice(c->device));
  return sf_nn_row_minima_dev(c, d_row_min, d_row_arg, d_status);
}

extern "C" int sf_nn_walk_device(sf_handle c, const double* d_row_min, const int32_t* d_row_arg, const int32_t* d_status,
                                 int32_t n_local, int32_t n_received, sf_match* d_matches, int32_t cap, int32_t* d_n_matches) {
  if (!c || !d_row_min || !d_row_arg || !d_n_matches || cap < 0 || (cap > 0 && !d_matches)) return SF_EINVAL;
  if (n_local <= 0 || n_received <= 0) return sf_fail(c, SF_EINVAL, "sf_nn_walk_device over %d x %d minima", n_local, n_received);
  SF_HIP(c, hipSetDevice(c->device));
  return sf_nn_walk_dev(c, d_row_min, d_row_arg, d_status, n_local, n_received, c->params.netvlad_distance,
                        c->params.netvlad_max_matches_nb, cap, nullptr, nullptr, d_matches, d_n_matches, nullptr);
}

extern "C" int sf_nn_walk(sf_handle c, const double* row_min, const int32_t* row_arg, int32_t n_local, int32_t n_received,
                          sf_match* out, int32_t cap, int32_t* n_out) {
  if (!c || !n_out || n_local < 0 || n_received < 0 || cap < 0 || (cap > 0 && !out) || (n_local > 0 && (!row_min || !row_arg)))
    return SF_EINVAL;
  *n_out = 0;
  if (n_local == 0 || n_received == 0) return SF_OK;
  return sf_nn_walk_host(c, row_min, row_arg, n_local, n_received, c->params.netvlad_distance,
                         c->params.netvlad_max_matches_nb, out, cap, n_out);
}

extern "C" int sf_nn_last_filter_dims(sf_handle c, int32_t* dims) {
  if (!c || !dims) return SF_EINVAL;
  *dims = c->nn_last_kdims;
  return SF_OK;
}

extern "C" int sf_nn_last_row_minima(sf_handle c, double* dist, int32_t* idx, int32_t cap) {
  if (!c) return SF_EINVAL;
  const int n = std::min<int>(cap, (int)c->last_row_min.size());
  for (int i = 0; i < n; ++i) {
    if (dist) dist[i] = c->last_row_min[i];
    if (idx) idx[i] = c->last_row_arg[i];
  }
  return SF_OK;
}
