// sf_comm.hip -- the one data-path collective of the separator finder: an RCCL all-gather of
// fixed-size separator records (one row of ReceiveSeparators.srv each) over xGMI, for C/C++ hosts
// that do not go through torch.distributed.  In the reference the same records travel as ROS1
// service requests between robots (ros_ws/src/multi_robot_separators/src/communication.cpp:15-52).
//
// RCCL is resolved lazily with dlopen so that single-GPU users carry no dependency and a process
// that already has an RCCL loaded (e.g. through PyTorch) keeps exactly one copy.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <vector>

#include "sf_internal.hpp"

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl() {
  static RcclApi api;
  if (api.lib) return &api;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {   // prefer a copy that is already in the process
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) return nullptr;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) return nullptr;
  api.lib = h;
  return &api;
}

int fail_rccl(sf_context* c, const char* what, ncclResult_t r) {
  RcclApi* a = rccl();
  return sf_fail(c, SF_ERCCL, "%s -> %s", what, (a && a->GetErrorString) ? a->GetErrorString(r) : "RCCL error");
}

}  // namespace

extern "C" int sf_comm_unique_id(uint8_t* out, int32_t cap) {
  if (!out || cap < SF_COMM_ID_BYTES) return SF_EINVAL;
  RcclApi* a = rccl();
  if (!a) return SF_ERCCL;
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return SF_ERCCL;
  static_assert(sizeof(id) == SF_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(out, &id, sizeof(id));
  return SF_OK;
}

extern "C" int sf_comm_init(sf_handle c, const uint8_t* unique_id, int32_t rank, int32_t world) {
  if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return SF_EINVAL;
  RcclApi* a = rccl();
  if (!a) return sf_fail(c, SF_ERCCL, "librccl.so could not be loaded");
  if (c->comm) return sf_fail(c, SF_EINVAL, "communicator already initialised");
  SF_HIP(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  ncclResult_t r = a->CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail_rccl(c, "ncclCommInitRank", r);
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return SF_OK;
}

extern "C" int sf_comm_destroy(sf_handle c) {
  if (!c) return SF_EINVAL;
  if (c->comm) {
    RcclApi* a = rccl();
    (void)hipStreamSynchronize(c->stream);
    if (a) (void)a->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  return SF_OK;
}

// ONE collective, no host involvement: every rank contributes cap_per_rank + 1 record slots -- slot 0 is a header whose
// first int32 is the rank's record count, stamped ON THE DEVICE by whoever produced the records (e.g. a compaction
// kernel writing its count there), slots 1.. are the records -- and receives world such blocks.  Asynchronous on the
// handle's stream: the caller reads the world counts out of the gathered headers behind its own synchronisation.
// This is the form the torch-side exchange of bench.py uses (dist.RecordExchange); a sub-millisecond step cannot
// afford the count round trip of a two-phase exchange.
extern "C" int sf_allgather_separators_device(sf_handle c, const sf_separator* d_send, sf_separator* d_all,
                                              int32_t cap_per_rank) {
  if (!c || !d_send || !d_all || cap_per_rank < 1) return SF_EINVAL;
  if (!c->comm) return sf_fail(c, SF_EINVAL, "sf_comm_init has not been called");
  RcclApi* a = rccl();
  SF_HIP(c, hipSetDevice(c->device));
  const ncclResult_t r = a->AllGather(d_send, d_all, (size_t)(cap_per_rank + 1) * sizeof(sf_separator), ncclUint8,
                                      (ncclComm_t)c->comm, c->stream);
  if (r != ncclSuccess) return fail_rccl(c, "ncclAllGather(records)", r);
  return SF_OK;
}

extern "C" int sf_allgather_bytes_device(sf_handle c, const void* d_send, void* d_all, size_t bytes_per_rank) {
  if (!c || !d_send || !d_all || bytes_per_rank == 0) return SF_EINVAL;
  if (!c->comm) return sf_fail(c, SF_EINVAL, "sf_comm_init has not been called");
  RcclApi* a = rccl();
  SF_HIP(c, hipSetDevice(c->device));
  const ncclResult_t r = a->AllGather(d_send, d_all, bytes_per_rank, ncclUint8, (ncclComm_t)c->comm, c->stream);
  if (r != ncclSuccess) return fail_rccl(c, "ncclAllGather(bytes)", r);
  return SF_OK;
}

// Host-count convenience form (synchronous): d_local: n_local records in device memory; d_all: world * cap_per_rank
// records in device memory (rank r's records start at r * cap_per_rank); counts: world entries on the host.  Also ONE
// collective: the count travels in the header slot of the block (uploaded with the records' staging copy), the
// gathered blocks are unpacked into the caller's layout with one pitched copy.
extern "C" int sf_allgather_separators(sf_handle c, const sf_separator* d_local, int32_t n_local,
                                       sf_separator* d_all, int32_t cap_per_rank, int32_t* counts) {
  if (!c || n_local < 0 || cap_per_rank < 1 || !d_all || !counts || (n_local > 0 && !d_local)) return SF_EINVAL;
  if (!c->comm) return sf_fail(c, SF_EINVAL, "sf_comm_init has not been called");
  if (n_local > cap_per_rank) return sf_fail(c, SF_ERANGE, "%d local records exceed the per-rank capacity %d", n_local, cap_per_rank);
  SF_HIP(c, hipSetDevice(c->device));
  const int world = c->comm_world;
  const size_t rec = sizeof(sf_separator), blk = (size_t)(cap_per_rank + 1) * rec;
  int rc;
  if ((rc = sf_buf_reserve(c, c->comm_scratch, blk * (size_t)(world + 1))) != SF_OK) return rc;
  char* d_send = (char*)c->comm_scratch.p;
  char* d_recv = d_send + blk;
  SF_HIP(c, hipMemsetAsync(d_send, 0, blk, c->stream));
  c->comm_count_host = n_local;       // (pageable source of a 4-byte copy: staged by the runtime before it returns)
  SF_HIP(c, hipMemcpyAsync(d_send, &c->comm_count_host, 4, hipMemcpyHostToDevice, c->stream));
  if (n_local)
    SF_HIP(c, hipMemcpyAsync(d_send + rec, d_local, (size_t)n_local * rec, hipMemcpyDeviceToDevice, c->stream));
  if ((rc = sf_allgather_separators_device(c, (const sf_separator*)d_send, (sf_separator*)d_recv, cap_per_rank)) != SF_OK)
    return rc;
  // unpack: records of rank r (block r, slots 1..) -> d_all + r * cap_per_rank; headers -> counts
  SF_HIP(c, hipMemcpy2DAsync(d_all, (size_t)cap_per_rank * rec, d_recv + rec, blk, (size_t)cap_per_rank * rec, world,
                             hipMemcpyDeviceToDevice, c->stream));
  SF_HIP(c, hipMemcpy2DAsync(counts, 4, d_recv, blk, 4, world, hipMemcpyDeviceToHost, c->stream));
  SF_HIP(c, hipStreamSynchronize(c->stream));
  return SF_OK;
}
