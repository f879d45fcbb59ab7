// Effect of the per-head stride (cache capacity) on a 32-head x 512 KiB streaming read (developer microbenchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const char* k, const char* v, unsigned* out, size_t head_stride, int nsplit) {
  const int tid = threadIdx.x, r = tid >> 3, c = tid & 7;
  const int split = blockIdx.x % nsplit, head = blockIdx.x / nsplit;
  const size_t base = (size_t)head * head_stride + (size_t)split * 128 * 256;
  const char* kb = k + base; const char* vb = v + base;
  u32x4 a[16];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t row = (size_t)(u * 32 + r) * 256;
    a[4 * u + 0] = *(const u32x4*)(kb + row + c * 16);
    a[4 * u + 1] = *(const u32x4*)(kb + row + 128 + c * 16);
    a[4 * u + 2] = *(const u32x4*)(vb + row + c * 16);
    a[4 * u + 3] = *(const u32x4*)(vb + row + 128 + c * 16);
  }
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) x ^= a[i][0] ^ a[i][1] ^ a[i][2] ^ a[i][3];
  if (x == 0x12345678u) out[0] = x;
}
int main() {
  const int L = 32, H = 32;
  unsigned* out; (void)hipMalloc(&out, 64);
  for (int rows : {2048, 4096}) {
    for (size_t extra_rows : {0, 16, 64, 128, 256, 512, 1024, 2048}) {
      const size_t hs = (size_t)(rows + extra_rows) * 256, bytes = hs * H;
      std::vector<char*> K(L), V(L);
      for (int l = 0; l < L; ++l) { (void)hipMalloc(&K[l], bytes); (void)hipMalloc(&V[l], bytes); (void)hipMemset(K[l], 1, bytes); (void)hipMemset(V[l], 1, bytes); }
      const int ns = rows / 128;
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      auto go = [&]() { for (int l = 0; l < L; ++l) hipLaunchKernelGGL(rd, dim3(H * ns), dim3(256), 0, 0, K[l], V[l], out, hs, ns); };
      go(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) go();
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const float us = ms * 1e3f / (10 * L);
      printf("rows=%d cap=%zu (head stride %zu KiB): %.2f us  %.2f TB/s\n", rows, rows + extra_rows, hs >> 10, us, 2.0 * H * rows * 256 / us / 1e6);
      for (int l = 0; l < L; ++l) { (void)hipFree(K[l]); (void)hipFree(V[l]); }
    }
  }
  return 0;
}
