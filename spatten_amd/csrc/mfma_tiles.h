// mfma_tiles.h — the matrix-core and LDS-DMA primitives shared by the flash prefill kernels (prefill_attn.hip) and the grouped-query
// decode kernel (decode_gqa.hip): v_mfma_f32_32x32x16 on 16-bit operands, and 16-byte-per-lane global -> LDS loads through a
// buffer descriptor (gfx950: buffer_load_dwordx4 ... lds).
#pragma once
#include "common.h"

namespace spatten {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Operand layout of the 32x32x16 forms (lane l: index l & 31, k-elements 8 (l >> 5) .. +8 of the 16); accumulator register r of
// lane l: row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  typedef bf16_t frag __attribute__((ext_vector_type(8)));
  __device__ static inline f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<f16_t> {
  typedef f16_t frag __attribute__((ext_vector_type(8)));
  __device__ static inline f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// buffer-descriptor LDS-DMA (device-only: the host pass of a __global__ template must not see the target builtins).
// Lane l fetches 16 bytes at base + soff + voff(l) into lds_dst + 16 l (lds_dst wave-uniform); bytes beyond `bytes`
// read as zero.  The descriptor is rebuilt from wave-uniform values at every call (a few SALU moves).
__device__ inline void dma16(const void* base, int64_t bytes, char* lds_dst, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0,
                                                                     (int)(bytes < 0x7FFFFFFF ? bytes : 0x7FFFFFFF), 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

}  // namespace spatten
