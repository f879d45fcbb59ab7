// Which access pattern / extra work costs what for a 32 MiB (16 MiB K + 16 MiB V) launch? — developer microbenchmark.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: contiguous 16 loads/lane from one buffer (K then V halves)
// MODE 1: decode pattern: lane (r=tid/8, c=tid%8) loads row (u*32+r): bytes [16c,16c+16) and [128+16c, ...) from K and V
// MODE 2: like 1 but lane owns 32 contiguous bytes of the row (c*32) -> two adjacent 16-B loads
// MODE 3: like 1 + block barrier + ds_bpermute reduction chain (simulated softmax dependency)
template <int MODE>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ k, const char* __restrict__ v, unsigned* out) {
  const int tid = threadIdx.x;
  u32x4 a[16];
  if (MODE == 0) {
    const u32x4* p = (const u32x4*)k + (size_t)blockIdx.x * 8 * 256 + tid;
    const u32x4* q = (const u32x4*)v + (size_t)blockIdx.x * 8 * 256 + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = p[i * 256]; a[8 + i] = q[i * 256]; }
  } else {
    const int r = tid >> 3, c = tid & 7;
    const char* kb = k + (size_t)blockIdx.x * 128 * 256;
    const char* vb = v + (size_t)blockIdx.x * 128 * 256;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t row = (size_t)(u * 32 + r) * 256;
      const int o0 = MODE == 2 ? c * 32 : c * 16, o1 = MODE == 2 ? c * 32 + 16 : 128 + c * 16;
      a[4 * u + 0] = *(const u32x4*)(kb + row + o0);
      a[4 * u + 1] = *(const u32x4*)(kb + row + o1);
      a[4 * u + 2] = *(const u32x4*)(vb + row + o0);
      a[4 * u + 3] = *(const u32x4*)(vb + row + o1);
    }
  }
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) x ^= a[i][0] ^ a[i][1] ^ a[i][2] ^ a[i][3];
  if (MODE == 3) {
    __shared__ unsigned s[4];
    for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if ((tid & 63) == 0) s[tid >> 6] = x;
    __syncthreads();
    x = s[0] ^ s[1] ^ s[2] ^ s[3];
    for (int off = 8; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
  }
  if (x == 0x12345678u) out[0] = x;
}
template <int MODE>
float run(std::vector<char*>& K, std::vector<char*>& V, unsigned* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t l = 0; l < K.size(); ++l) hipLaunchKernelGGL(rd<MODE>, dim3(512), dim3(256), 0, 0, K[l], V[l], out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r)
    for (size_t l = 0; l < K.size(); ++l) hipLaunchKernelGGL(rd<MODE>, dim3(512), dim3(256), 0, 0, K[l], V[l], out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (10 * K.size());
}
int main() {
  const int L = 32; const size_t bytes = 16 << 20;
  unsigned* out; (void)hipMalloc(&out, 64);
  std::vector<char*> K(L), V(L);
  for (int l = 0; l < L; ++l) { (void)hipMalloc(&K[l], bytes + 4096); (void)hipMalloc(&V[l], bytes + 4096); (void)hipMemset(K[l], 1, bytes); (void)hipMemset(V[l], 1, bytes); }
  printf("mode0 contiguous : %.2f us\n", run<0>(K, V, out));
  printf("mode1 lo/hi split: %.2f us\n", run<1>(K, V, out));
  printf("mode2 32B/lane   : %.2f us\n", run<2>(K, V, out));
  printf("mode3 +sync+shfl : %.2f us\n", run<3>(K, V, out));
  return 0;
}
