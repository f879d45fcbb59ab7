// Where do workgroups land, and how fast does a write-through store become visible to a poller on the same / another
// XCD?   hipcc --offload-arch=gfx950 -O2 tools/mb/xcc_map.cpp -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

__global__ void where(unsigned* out) {
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (threadIdx.x == 0) out[lin] = xcc_id();
}

// workgroup `writer` waits, then publishes; every other workgroup polls (mode 0: L2-scope loads, 1: agent scope, 2: L2
// scope for 64 polls then agent scope) and records when it saw the value (100 MHz wall clock) and its XCC
__global__ void pingpong(unsigned long long* flag, unsigned long long* t_seen, unsigned* xcc, unsigned long long* t_pub,
                         int writer, int mode, unsigned long long tag) {
  const int w = blockIdx.x;
  if (threadIdx.x != 0) return;
  xcc[w] = xcc_id();
  if (w == writer) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 3000) {}     // 30 us: everybody is polling by now
    *t_pub = wall_clock64();
    if (mode == 4) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  unsigned long long v = 0;
  long spins = 0;
  while (true) {
    if (mode >= 3) {            // 3: L1 invalidate + L2-scope load (writer: write-through store); 4: same, writer: plain store
      asm volatile("buffer_inv sc0" ::: "memory");
      v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode == 0 || (mode == 2 && spins < 64)) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == tag) break;
    if (++spins > 200000) break;
  }
  t_seen[w] = (v == tag) ? wall_clock64() : 0ull;
}

int main() {
  unsigned* d; hipMalloc(&d, 4096 * 4);
  for (auto g : {dim3(64, 1, 1), dim3(8, 8, 1), dim3(16, 2, 2)}) {
    hipLaunchKernelGGL(where, g, dim3(64), 0, 0, d);
    std::vector<unsigned> h(g.x * g.y * g.z);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("grid (%u,%u,%u): xcc by linear id:", g.x, g.y, g.z);
    for (size_t i = 0; i < h.size(); ++i) printf(" %u", h[i]);
    printf("\n");
  }
  const int W = 16;
  unsigned long long *flag, *seen, *pub; unsigned* xcc;
  hipMalloc(&flag, 256); hipMalloc(&seen, W * 8); hipMalloc(&pub, 8); hipMalloc(&xcc, W * 4);
  unsigned long long tag = 1;
  for (int mode = 0; mode < 5; ++mode)
    for (int rep = 0; rep < 3; ++rep, ++tag) {
      hipMemset(seen, 0, W * 8);
      hipLaunchKernelGGL(pingpong, dim3(W), dim3(64), 0, 0, flag, seen, xcc, pub, 8, mode, tag);
      hipDeviceSynchronize();
      std::vector<unsigned long long> hs(W); std::vector<unsigned> hx(W); unsigned long long hp;
      hipMemcpy(hs.data(), seen, W * 8, hipMemcpyDeviceToHost);
      hipMemcpy(hx.data(), xcc, W * 4, hipMemcpyDeviceToHost);
      hipMemcpy(&hp, pub, 8, hipMemcpyDeviceToHost);
      printf("mode %d (writer wg 8 on xcc %u): ", mode, hx[8]);
      for (int w = 0; w < W; ++w)
        if (w != 8) printf(" wg%d/x%u:%s%lld", w, hx[w], hs[w] ? "" : "never", hs[w] ? (long long)(hs[w] - hp) * 10 : 0ll);
      printf("  (ns after the store was issued)\n");
    }
  return 0;
}
