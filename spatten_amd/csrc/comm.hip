// comm.hip — the one exchange step of the head-parallel path behind the C ABI: all-gather of the per-rank
// attention outputs [B, q, H/G * d] over RCCL on a communicator the LIBRARY owns, so the collective is an ordinary
// stream operation: it can be captured into the per-token HIP graph next to the attention launches (torch's process
// group cannot: its watchdog aborts on captured work on this stack), and a non-Python host can drive it.
//
// RCCL is loaded lazily (dlopen) the first time a communicator is asked for: a single-GPU user of libspatten_hip.so
// never pays for, or depends on, librccl.
#include <dlfcn.h>
#include <stdlib.h>

#include "common.h"

namespace spatten {

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // ncclUniqueId is passed BY VALUE: see init below
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  bool ok = false;
};

struct UniqueId { char internal[SPATTEN_COMM_ID_BYTES]; };   // = ncclUniqueId (NCCL_UNIQUE_ID_BYTES 128)

static RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // SPATTEN_RCCL_LIB: an explicit library path (deployment with a private RCCL build; also how the tests reach the
    // "RCCL not installed" branch)
    const char* forced = getenv("SPATTEN_RCCL_LIB");
    if (forced && *forced) {
      api.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    } else {
      const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
      for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
      }
    }
    if (api.handle) {
      api.GetUniqueId = (int (*)(void*))dlsym(api.handle, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(void**, int, const void*, int))dlsym(api.handle, "ncclCommInitRank");
      api.CommDestroy = (int (*)(void*))dlsym(api.handle, "ncclCommDestroy");
      api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.handle, "ncclAllGather");
      api.CommCount = (int (*)(void*, int*))dlsym(api.handle, "ncclCommCount");
      api.CommUserRank = (int (*)(void*, int*))dlsym(api.handle, "ncclCommUserRank");
      api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
    }
  }
  return api;
}

}  // namespace spatten

using namespace spatten;

extern "C" int spatten_comm_unique_id(void* id_out) {
  if (!id_out) return SPATTEN_ERR_INVALID;
  RcclApi& r = rccl();
  if (!r.ok) return SPATTEN_ERR_UNSUPPORTED;
  return r.GetUniqueId(id_out) == 0 ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_comm_init(void** comm_out, int rank, int nranks, const void* unique_id) {
  if (!comm_out || !unique_id || nranks <= 0 || rank < 0 || rank >= nranks) return SPATTEN_ERR_INVALID;
  RcclApi& r = rccl();
  if (!r.ok) return SPATTEN_ERR_UNSUPPORTED;
  // ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId /* by value, 128 bytes */, int rank): call it through
  // its real prototype
  typedef int (*init_fn)(void**, int, UniqueId, int);
  UniqueId id;
  __builtin_memcpy(&id, unique_id, sizeof(id));
  void* comm = nullptr;
  const int rc = ((init_fn)(void*)r.CommInitRank)(&comm, nranks, id, rank);
  if (rc != 0 || !comm) return SPATTEN_ERR_LAUNCH;
  *comm_out = comm;
  return SPATTEN_OK;
}

extern "C" int spatten_comm_destroy(void* comm) {
  if (!comm) return SPATTEN_OK;
  RcclApi& r = rccl();
  if (!r.ok) return SPATTEN_ERR_UNSUPPORTED;
  return r.CommDestroy(comm) == 0 ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_comm_info(void* comm, int* nranks_out, int* rank_out) {
  if (!comm || !nranks_out || !rank_out) return SPATTEN_ERR_INVALID;
  RcclApi& r = rccl();
  if (!r.ok || !r.CommCount || !r.CommUserRank) return SPATTEN_ERR_UNSUPPORTED;
  return (r.CommCount(comm, nranks_out) == 0 && r.CommUserRank(comm, rank_out) == 0) ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!comm || !send || !recv) return SPATTEN_ERR_INVALID;
  if (bytes_per_rank == 0) return SPATTEN_OK;
  RcclApi& r = rccl();
  if (!r.ok) return SPATTEN_ERR_UNSUPPORTED;
  // bytes as ncclChar (= 0): the payload is opaque to the collective
  return r.AllGather(send, recv, bytes_per_rank, 0, comm, (hipStream_t)stream) == 0 ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}
