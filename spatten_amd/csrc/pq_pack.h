// pq_pack.h — packing one row of the profiled progressive-quantisation planes (include/spatten.h "Bit profiles and the quantised VALUE
// plane"; oracle: pq_quantize / pq_quantize_values).  Shared by pq.hip (the pack kernels, the one-launch append) and pq_decode.hip
// (the MSB pass that appends the step's row itself).
#pragma once
#include "common.h"

namespace spatten {

// 16 unsigned fields of BITS bits (piece order t = 0..15) -> the piece's dwords: BITS 8 -> 4, 6 -> 3, 4 -> 2
template <int BITS>
__device__ inline void pack_fields(const uint32_t (&f)[16], uint32_t (&w)[BITS / 2]) {
#pragma unroll
  for (int i = 0; i < BITS / 2; ++i) w[i] = 0u;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int bit = BITS * t, wi = bit / 32, off = bit % 32;
    w[wi] |= f[t] << off;
    if (off + BITS > 32) w[wi + 1] |= f[t] >> (32 - off);
  }
}
template <int W> __device__ inline void store_words(uint8_t* p, const uint32_t (&w)[W]) {
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < W; ++i) q[i] = w[i];
}

// symmetric per-row quantisation of a lane's 16 elements to `bits` bits: the row maximum over the LPR lanes of the row, scale =
// max / (2^(bits-1) - 1), round to nearest even, clamp (oracle: pq_quantize / pq_quantize_values)
template <int LPR>
__device__ inline void quantise_piece(const float (&x)[16], int bits, float& sc_out, int (&q)[16]) {
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(x[e]));
  amax = fmaxf(amax, dpp_mov<kDppXor1>(amax));
  amax = fmaxf(amax, dpp_mov<kDppXor2>(amax));
  if (LPR == 8) amax = fmaxf(amax, dpp_mov<kDppHalfMirror>(amax));
  const float qmax = (float)((1 << (bits - 1)) - 1);
  const float sc = amax > 0.f ? amax / qmax : 1.0f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    int qv = (int)rintf(x[e] / sc);
    q[e] = max(-(int)qmax - 1, min((int)qmax, qv));
  }
  sc_out = sc;
}
// one row's pieces of the profiled planes from the lane's 16 rotated-key and 16 value elements (piece order): the packed words
template <int D, int KB, int VB>
struct PlanePieces { uint32_t wm[KB / 2], wl[2], wv[VB / 2]; float ks, vs; };
template <int D, int KB, int VB>
__device__ inline void pack_plane_pieces(const float (&kx)[16], const float (&vx)[16], PlanePieces<D, KB, VB>& o) {
  constexpr int LPR = D / 16;
  {   // keys: T = KB + 4 bits, MSB plane = q >> 4, LSB plane = q & 15
    int q[16];
    quantise_piece<LPR>(kx, KB + 4, o.ks, q);
    uint32_t fm[16], fl[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { fm[e] = (uint32_t)((q[e] >> 4) + (1 << (KB - 1))); fl[e] = (uint32_t)(q[e] & 15); }
    pack_fields<KB>(fm, o.wm);
    pack_fields<4>(fl, o.wl);
  }
  {   // values: one plane of VB bits
    int q[16];
    quantise_piece<LPR>(vx, VB, o.vs, q);
    uint32_t f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) f[e] = (uint32_t)(q[e] + (1 << (VB - 1)));
    pack_fields<VB>(f, o.wv);
  }
}
template <int D, int KB, int VB>
__device__ inline void store_plane_words(const PlanesDev& pl, int b, int hkv, int row, int c, const PlanePieces<D, KB, VB>& o) {
  const int64_t so = b * pl.sc_sb + hkv * pl.sc_sh + row;
  store_words<KB / 2>(pl.km + b * pl.km_sb + hkv * pl.km_sh + (int64_t)row * (D * KB / 8) + 2 * KB * c, o.wm);
  store_words<2>(pl.kl + b * pl.kl_sb + hkv * pl.kl_sh + (int64_t)row * (D / 2) + 8 * c, o.wl);
  store_words<VB / 2>(pl.vq + b * pl.vq_sb + hkv * pl.vq_sh + (int64_t)row * (D * VB / 8) + 2 * VB * c, o.wv);
  if (c == 0) { pl.ks[so] = o.ks; pl.vs[so] = o.vs; }
}
template <int D, int KB, int VB>
__device__ inline void store_plane_pieces(const PlanesDev& pl, int b, int hkv, int row, int c, const float (&kx)[16], const float (&vx)[16]) {
  PlanePieces<D, KB, VB> o;
  pack_plane_pieces<D, KB, VB>(kx, vx, o);
  store_plane_words<D, KB, VB>(pl, b, hkv, row, c, o);
}

}  // namespace spatten
