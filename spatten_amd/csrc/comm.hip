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

// ------------------------------------------------------------------------------------------------
// Peer-store all-gather (ABI 4; SURVEY §8e: "decode messages are latency-bound -> single-shot direct writes to all 7 peers (one
// link each, all links concurrently) instead of ring").  The decode step's exchange is 1 KiB per rank per layer at 8 GPUs: a
// collective library's protocol (channels, proxy, ring / tree steps) is all latency there.  Here every rank owns a RECEIVE
// WINDOW in its own HBM, mapped into every peer through hipIpc; one launch per all-gather:
//   block p of rank r   writes r's slice into slot r of peer p's window (xGMI stores straight into the peer's memory), fences,
//                       publishes the epoch into flag r of that window; then polls flag p of its OWN window for the same epoch
//                       and copies slot p into the caller's receive buffer.
// Two window halves alternate by epoch parity: a peer can only overwrite half e & 1 at epoch e + 2, after it has seen this
// rank's flag of epoch e + 1 — which this rank publishes only after its epoch-e launch (copy-out included) has completed.
// The epoch lives in device memory and is advanced by the launch itself, so a captured graph replays it.  Every word that
// crosses devices is written and read with system-scope accesses (the window is fine-grained memory where the runtime offers
// it).  A flag that does not arrive within the bounded spin sets the peer object's error word (spatten_peer_status).
// Messages up to the window's slot size (spatten_peer_create); the prefill exchange (MiBs per rank) stays with RCCL.
// ------------------------------------------------------------------------------------------------
namespace spatten {

constexpr int kPeerMaxRanks = 16;
constexpr size_t kPeerFlagBytes = 4096;      // 2 halves x 16 ranks x 64-byte lines

struct PeerStore {
  int rank = 0, nranks = 1;
  size_t slot = 0;                 // bytes per rank per half
  char* window = nullptr;          // own window: [flags 4 KiB][2][nranks][slot]
  char* peers[kPeerMaxRanks] = {}; // every rank's window as mapped HERE (peers[rank] = window)
  bool opened[kPeerMaxRanks] = {};
  unsigned* state = nullptr;       // device: {epoch, blocks done, error}
  bool finegrained = false;
};

struct PeerArgs {
  char* peers[kPeerMaxRanks];
  const char* send; char* recv; unsigned* state;
  size_t bytes, slot; int rank, nranks;
};

__global__ __launch_bounds__(256) void peer_allgather_kernel(const PeerArgs a) {
  using u64 = unsigned long long;
  const int p = blockIdx.x, tid = threadIdx.x;
  const unsigned epoch = __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const unsigned half = epoch & 1u;
  const size_t words = a.bytes / 8;
  // 1. my slice -> slot `rank` of peer p's window
  {
    u64* dst = reinterpret_cast<u64*>(a.peers[p] + kPeerFlagBytes + ((size_t)half * a.nranks + a.rank) * a.slot);
    const u64* src = reinterpret_cast<const u64*>(a.send);
    for (size_t i = tid; i < words; i += blockDim.x) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the payload went out write-through (system-scope stores): every storing wave drains its own stores, then ONE lane
    // raises the flag — no cache-wide release fence (a system-scope fence writes back the whole L2: several us per launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(reinterpret_cast<unsigned*>(a.peers[p] + (half * kPeerMaxRanks + a.rank) * 64), epoch, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. peer p's slice out of MY window
  {
    char* mine = a.peers[a.rank];
    const unsigned* flag = reinterpret_cast<const unsigned*>(mine + (half * kPeerMaxRanks + p) * 64);
    __shared__ int ok;
    if (tid == 0) {
      int spins = 0;
      unsigned seen;
      do {
        seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (the slices are read with
        if (seen != epoch) __builtin_amdgcn_s_sleep(2);                                   //  system-scope loads: no acquire fence)
      } while (seen != epoch && ++spins < (1 << 22));
      ok = seen == epoch;
      if (!ok) atomicOr(a.state + 2, 1u);
    }
    __syncthreads();
    const u64* src = reinterpret_cast<const u64*>(mine + kPeerFlagBytes + ((size_t)half * a.nranks + p) * a.slot);
    u64* dst = reinterpret_cast<u64*>(a.recv + (size_t)p * a.bytes);
    for (size_t i = tid; i < words; i += blockDim.x)
      dst[i] = ok ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
  }
  // 3. the last block of the launch advances the epoch (the launch is its own epoch source: replayable)
  __syncthreads();
  if (tid == 0) {
    const unsigned done = atomicAdd(a.state + 1, 1u);
    if (done == (unsigned)a.nranks - 1u) {
      __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.state, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace spatten

extern "C" int spatten_peer_create(void** peer_out, int rank, int nranks, size_t max_bytes_per_rank, void* handle_out) {
  if (!peer_out || !handle_out || nranks <= 0 || nranks > kPeerMaxRanks || rank < 0 || rank >= nranks || max_bytes_per_rank == 0)
    return SPATTEN_ERR_INVALID;
  static_assert(sizeof(hipIpcMemHandle_t) <= SPATTEN_PEER_HANDLE_BYTES, "handle size");
  PeerStore* ps = new PeerStore();
  ps->rank = rank; ps->nranks = nranks;
  ps->slot = (max_bytes_per_rank + 255) / 256 * 256;
  const size_t bytes = kPeerFlagBytes + 2 * (size_t)nranks * ps->slot;
  void* w = nullptr;
  // The exchange's relaxed system-scope stores + flag are only a protocol on FINE-GRAINED memory (a peer's xGMI writes reach a
  // kernel that is already polling); a coarse-grained window would need release / acquire fences around every slice.  Several
  // ranks without fine-grained memory: unsupported — the caller keeps RCCL.  One rank (loopback) has no peer.
  if (hipExtMallocWithFlags(&w, bytes, hipDeviceMallocFinegrained) == hipSuccess && w) ps->finegrained = true;
  else {
    (void)hipGetLastError();
    if (nranks > 1) { delete ps; return SPATTEN_ERR_UNSUPPORTED; }
    if (hipMalloc(&w, bytes) != hipSuccess || !w) { delete ps; return SPATTEN_ERR_LAUNCH; }
  }
  ps->window = (char*)w;
  void* st = nullptr;
  if (hipMemset(w, 0, bytes) != hipSuccess || hipMalloc(&st, 256) != hipSuccess || hipMemset(st, 0, 256) != hipSuccess) {
    (void)hipFree(w); if (st) (void)hipFree(st); delete ps; return SPATTEN_ERR_LAUNCH;
  }
  ps->state = (unsigned*)st;
  ps->peers[rank] = ps->window;
  __builtin_memset(handle_out, 0, SPATTEN_PEER_HANDLE_BYTES);
  if (nranks > 1) {
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, w) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); (void)hipFree(st); delete ps; return SPATTEN_ERR_UNSUPPORTED; }
    __builtin_memcpy(handle_out, &h, sizeof(h));
  }
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(w); (void)hipFree(st); delete ps; return SPATTEN_ERR_LAUNCH; }
  *peer_out = ps;
  return SPATTEN_OK;
}

extern "C" int spatten_peer_connect(void* peer, const void* handles) {
  PeerStore* ps = (PeerStore*)peer;
  if (!ps || (ps->nranks > 1 && !handles)) return SPATTEN_ERR_INVALID;
  for (int r = 0; r < ps->nranks; ++r) {
    if (r == ps->rank || ps->peers[r]) continue;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, (const char*)handles + (size_t)r * SPATTEN_PEER_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) { (void)hipGetLastError(); return SPATTEN_ERR_LAUNCH; }
    ps->peers[r] = (char*)p;
    ps->opened[r] = true;
  }
  return SPATTEN_OK;
}

extern "C" int spatten_peer_allgather(void* peer, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  PeerStore* ps = (PeerStore*)peer;
  if (!ps || !send || !recv) return SPATTEN_ERR_INVALID;
  if (bytes_per_rank == 0) return SPATTEN_OK;
  if (bytes_per_rank > ps->slot || bytes_per_rank % 8 != 0) return SPATTEN_ERR_UNSUPPORTED;
  PeerArgs a;
  for (int r = 0; r < kPeerMaxRanks; ++r) a.peers[r] = r < ps->nranks ? ps->peers[r] : nullptr;
  for (int r = 0; r < ps->nranks; ++r) if (!a.peers[r]) return SPATTEN_ERR_INVALID;      // spatten_peer_connect first
  a.send = (const char*)send; a.recv = (char*)recv; a.state = ps->state;
  a.bytes = bytes_per_rank; a.slot = ps->slot; a.rank = ps->rank; a.nranks = ps->nranks;
  hipLaunchKernelGGL(peer_allgather_kernel, dim3((unsigned)ps->nranks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_peer_status(void* peer, void* stream) {
  PeerStore* ps = (PeerStore*)peer;
  if (!ps) return SPATTEN_ERR_INVALID;
  unsigned st[3] = {0, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(st, ps->state, sizeof(st), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return SPATTEN_ERR_LAUNCH;
  if (st[2] == 0) return SPATTEN_OK;
  (void)hipMemsetAsync(ps->state + 2, 0, sizeof(unsigned), s);
  return SPATTEN_ERR_TIMEOUT;
}

extern "C" int spatten_peer_destroy(void* peer) {
  PeerStore* ps = (PeerStore*)peer;
  if (!ps) return SPATTEN_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < ps->nranks; ++r) if (ps->opened[r]) (void)hipIpcCloseMemHandle(ps->peers[r]);
  if (ps->window) (void)hipFree(ps->window);
  if (ps->state) (void)hipFree(ps->state);
  delete ps;
  return SPATTEN_OK;
}
