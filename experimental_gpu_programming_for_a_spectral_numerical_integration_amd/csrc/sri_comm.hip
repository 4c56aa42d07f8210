// sri_comm.hip — the ONE exchange step of the multi-GPU path (SURVEY.md §8e, §8b `sri_allgather`): reassembling the
// integrated state of a rod batch that was sharded by columns over the GPUs of one node.  RCCL over xGMI, called from
// C++ on the caller's HIP stream, so that a C++ host program (the reference's main.cpp is C++) needs neither Python nor
// torch.distributed.  The apply itself never communicates: rods are independent and the operator is replicated.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on first use): single-GPU users of libsri_hip.so do not load the
// 570 MB collective library, and a process that already holds RCCL (PyTorch's bundled copy) shares that instance.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "sri_internal.h"

// RCCL is bound at run time, so only its TYPES are needed at compile time.  A ROCm installation without the RCCL
// development headers must still be able to build libsri_hip.so — the single-GPU apply does not need them: the few
// declarations used here are then made locally (ABI of rccl.h, stable since NCCL 2.x).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
}
#endif

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

std::once_flag g_once;
Rccl g_rccl;
thread_local int g_last_rccl = 0;

template <typename F>
bool bind(void* so, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(so, name));
  return fn != nullptr;
}

const Rccl& rccl() {
  std::call_once(g_once, [] {
    Rccl& r = g_rccl;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) return;
    r.ok = bind(r.so, "ncclGetUniqueId", r.GetUniqueId) && bind(r.so, "ncclCommInitRank", r.CommInitRank) &&
           bind(r.so, "ncclCommDestroy", r.CommDestroy) && bind(r.so, "ncclAllGather", r.AllGather) &&
           bind(r.so, "ncclBroadcast", r.Broadcast) && bind(r.so, "ncclGroupStart", r.GroupStart) &&
           bind(r.so, "ncclGroupEnd", r.GroupEnd) && bind(r.so, "ncclGetErrorString", r.GetErrorString);
  });
  return g_rccl;
}

int fail(ncclResult_t e) {
  g_last_rccl = static_cast<int>(e);
  return SRI_ECOMM;
}

// Makes `device` current for the scope of a call and restores the caller's device afterwards.
struct CommDeviceGuard {
  int prev = -1;
  bool switched = false;
  int enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != device) {
      e = hipSetDevice(device);
      switched = e == hipSuccess;
    }
    if (e != hipSuccess) {
      sri::record_hip_error(static_cast<int>(e));
      (void)hipGetLastError();
      return SRI_EHIP;
    }
    return SRI_OK;
  }
  ~CommDeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct sri_comm_handle {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  bool force_ragged = false;   // SRI_DEBUG=comm_force_ragged=1 at creation: even shards take the ragged route (lets a one-GPU box test it)
};

extern "C" {

int sri_comm_unique_id(void* id) {
  if (!id) return SRI_EINVAL;
  static_assert(SRI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "sri_capi.h must mirror RCCL's unique-id size");
  const Rccl& r = rccl();
  if (!r.ok) return SRI_EUNSUPPORTED;
  ncclUniqueId u;
  ncclResult_t e = r.GetUniqueId(&u);
  if (e != ncclSuccess) return fail(e);
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return SRI_OK;
}

int sri_comm_create(sri_comm_t** out, const void* id, int rank, int world, int device) {
  if (!out) return SRI_EINVAL;
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) return SRI_EINVAL;
  int count = 0;
  hipError_t he = hipGetDeviceCount(&count);
  if (he != hipSuccess || count <= 0 || device < 0 || device >= count) {
    if (he != hipSuccess) sri::record_hip_error(static_cast<int>(he));
    return SRI_ENODEVICE;
  }
  const Rccl& r = rccl();
  if (!r.ok) return SRI_EUNSUPPORTED;
  CommDeviceGuard guard;
  int rc = guard.enter(device);
  if (rc != SRI_OK) return rc;
  sri_comm_handle* h = new (std::nothrow) sri_comm_handle;
  if (!h) return SRI_ENOMEM;
  h->rank = rank;
  h->world = world;
  h->force_ragged = sri::debug_option("comm_force_ragged") == "1";
  h->device = device;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t e = r.CommInitRank(&h->comm, world, u, rank);     // collective over all `world` callers
  if (e != ncclSuccess) {
    delete h;
    return fail(e);
  }
  *out = h;
  return SRI_OK;
}

int sri_comm_destroy(sri_comm_t* h) {
  if (!h) return SRI_OK;
  int rc = SRI_OK;
  CommDeviceGuard guard;
  (void)guard.enter(h->device);
  if (h->comm) {
    ncclResult_t e = rccl().CommDestroy(h->comm);
    if (e != ncclSuccess) rc = fail(e);
  }
  delete h;
  return rc;
}

int sri_comm_rank(const sri_comm_t* h, int* rank, int* world) {
  if (!h) return SRI_EINVAL;
  if (rank) *rank = h->rank;
  if (world) *world = h->world;
  return SRI_OK;
}

// Gather `counts[r]` doubles from every rank r into d_all at offset sum(counts[0..r)).  counts == nullptr: every
// rank contributes `count_local` (one ncclAllGather).  Ragged shards (rods % world != 0, parallel.shard_columns) are
// one group of per-root broadcasts — RCCL fuses the group into one launch.
static int allgather_doubles(sri_comm_t* h, const double* d_local, size_t count_local, double* d_all, const size_t* counts,
                             void* stream) {
  if (!h || !h->comm || !d_all || (!d_local && count_local)) return SRI_EINVAL;
  CommDeviceGuard guard;
  int grc = guard.enter(h->device);
  if (grc != SRI_OK) return grc;
  const Rccl& r = rccl();
  hipStream_t s = static_cast<hipStream_t>(stream);
  bool equal = true;
  if (counts) {
    if (counts[h->rank] != count_local) return SRI_EINVAL;
    for (int i = 0; i < h->world; ++i) equal = equal && counts[i] == count_local;
    if (h->force_ragged) equal = false;
  }
  ncclResult_t e;
  if (equal) {
    if (count_local == 0) return SRI_OK;
    e = r.AllGather(d_local, d_all, count_local, ncclDouble, h->comm, s);
    return e == ncclSuccess ? SRI_OK : fail(e);
  }
  e = r.GroupStart();
  if (e != ncclSuccess) return fail(e);
  size_t off = 0;
  ncclResult_t first_bad = ncclSuccess;
  for (int root = 0; root < h->world; ++root) {
    if (counts[root]) {
      e = r.Broadcast(root == h->rank ? d_local : d_all + off, d_all + off, counts[root], ncclDouble, root, h->comm, s);
      if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
    }
    off += counts[root];
  }
  e = r.GroupEnd();
  if (first_bad != ncclSuccess) return fail(first_bad);
  return e == ncclSuccess ? SRI_OK : fail(e);
}

int sri_allgather_tips(sri_comm_t* h, const double* d_tips_local, size_t columns_local, double* d_tips_all,
                       const size_t* columns_per_rank, void* hip_stream) {
  return allgather_doubles(h, d_tips_local, columns_local, d_tips_all, columns_per_rank, hip_stream);
}

int sri_allgather_panel(sri_comm_t* h, const double* dY_local, size_t rows, size_t columns_local, double* dY_all,
                        const size_t* columns_per_rank, void* hip_stream) {
  if (!h || rows == 0) return SRI_EINVAL;
  if (!columns_per_rank) return allgather_doubles(h, dY_local, rows * columns_local, dY_all, nullptr, hip_stream);
  try {                                                           // nothing may throw across the C ABI
    std::vector<size_t> counts(static_cast<size_t>(h->world));
    for (int i = 0; i < h->world; ++i) counts[static_cast<size_t>(i)] = rows * columns_per_rank[i];
    return allgather_doubles(h, dY_local, rows * columns_local, dY_all, counts.data(), hip_stream);
  } catch (const std::bad_alloc&) {
    return SRI_ENOMEM;
  }
}

int sri_last_rccl_error(void) { return g_last_rccl; }

const char* sri_rccl_strerror(int rccl_result) {
  const Rccl& r = rccl();
  return r.ok ? r.GetErrorString(static_cast<ncclResult_t>(rccl_result)) : "RCCL not loaded";
}

}  // extern "C"
