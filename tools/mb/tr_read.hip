#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* in, uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  // mode 0: 16 lanes of a group point at consecutive 8-byte chunks (a 4x16 row-major matrix of 16-bit elements per group)
  // mode 1: lane i points at row (i>>2) of a pitch-128-element matrix, columns 4*(i&3) (+ 16*g)
  int off = mode == 0 ? (g * 64 + li * 4) : ((li >> 2) * 128 + 4 * (li & 3) + 16 * g);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + off));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}
int main() {
  uint16_t h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (uint16_t)i;
  uint16_t *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(di, dout, mode);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
