// pq.hip — progressive quantisation of the key cache (SpAtten H4; PARITY UNPINNED: the reference's Python has no
// numeric implementation — restated from the RTL control flow and checked against oracle/spatten_oracle.py):
//   * K is stored as an MSB plane (4 bits / element) plus a separate LSB plane (4 bits / element) and a per-row
//     scale (MatrixFetcher.scala:48-51, 341-348; SpAttenController.scala:35-39): q8 = msb*16 + lsb,  x ~ q8 * scale
//   * pass 1 scores every key from the MSB plane only (MSBs left-aligned, LSBs zero)  -> softmax statistics
//   * need_lsb = max_j prob_j < threshold  (RequantDecision.scala:44-72); only then the LSB plane is fetched and
//     the row is recomputed ONCE (SpAttenController.scala:402)
//   * V is not progressive (SpAttenController.scala:723)
// The planes quantise the ROTATED keys (the shadow, see decode_attn.hip): scores are q_rot · k_rot.
// HBM bytes per key row (d = 128): 64 (MSB) [+ 64 (LSB) when refetched] + 4 (scale) + 256 (V, bf16) vs 512 dense.
#include "common.h"
#include "pq_pack.h"

namespace spatten {

// ---- pack: rows [lo, hi) of kr [B,Hkv,cap,D] -> msb/lsb [B,Hkv,cap,D/2] bytes, scale [B,Hkv,cap] fp32 ------------
// 16 lanes per row (8 elements each); symmetric per-row 8-bit quantiser: scale = amax/127, q = clip(rint(x/scale)).
template <typename T, int D>
__global__ __launch_bounds__(256) void pq_pack_kernel(const T* __restrict__ kr, int64_t kv_sb, int64_t kv_sh,
                                                      uint8_t* __restrict__ msb, uint8_t* __restrict__ lsb,
                                                      float* __restrict__ scale, int64_t pl_sb, int64_t pl_sh,
                                                      int64_t sc_sb, int64_t sc_sh, int lo, int hi) {
  constexpr int LPR = D / 8;
  constexpr int RPB = 256 / LPR;
  const int tid = threadIdx.x, c = tid % LPR, r = tid / LPR;
  const int row = lo + blockIdx.x * RPB + r;
  const int hkv = blockIdx.y, b = blockIdx.z;
  if (row >= hi) return;            // whole row-groups leave together (LPR divides the wave)
  float x[8];
  Vec8<T>::unpack(Vec8<T>::ldg(kr + b * kv_sb + hkv * kv_sh + (int64_t)row * D + 8 * c), x);
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(x[e]));
  amax = fmaxf(amax, dpp_mov<kDppXor1>(amax));
  amax = fmaxf(amax, dpp_mov<kDppXor2>(amax));
  amax = fmaxf(amax, dpp_mov<kDppHalfMirror>(amax));
  if (LPR == 16) amax = fmaxf(amax, dpp_mov<kDppMirror>(amax));
  const float sc = amax > 0.f ? amax / 127.0f : 1.0f;
  uint32_t m4 = 0, l4 = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int qv = (int)rintf(x[e] / sc);
    qv = max(-128, min(127, qv));
    m4 |= (uint32_t)((qv >> 4) & 15) << (4 * e);     // arithmetic shift: floor
    l4 |= (uint32_t)(qv & 15) << (4 * e);
  }
  const int64_t po = b * pl_sb + hkv * pl_sh + (int64_t)row * (D / 2) + 4 * c;
  *reinterpret_cast<uint32_t*>(msb + po) = m4;
  *reinterpret_cast<uint32_t*>(lsb + po) = l4;
  if (c == 0) scale[b * sc_sb + hkv * sc_sh + row] = sc;
}

// ---- one decode step's append in device-length form (step.hip): row n - 1 (n = state word 0, already advanced) of k / v,
// its rotation into the shadow with the state's staged rotary row 1 (modify_llama.py:95-104) and — when planes are given —
// that row's MSB / LSB nibbles and scale, bit for bit what kv_append_kernel + pq_pack_kernel leave there.  Nothing in the
// launch depends on a host length, so the progressive-quantisation decode step (whose attention launch appends nothing) can
// be captured into the per-token graph.  D/16 lanes per (b, h): 8 lower-half + 8 upper-half elements each.
template <typename T, int D>
__global__ __launch_bounds__(64) void kv_append_pq_step_kernel(const T* __restrict__ k_new, const T* __restrict__ v_new,
                                                               int64_t new_sb, int64_t new_sh, T* __restrict__ kc,
                                                               T* __restrict__ krc, T* __restrict__ vc, int64_t kv_sb,
                                                               int64_t kv_sh, uint8_t* __restrict__ msb,
                                                               uint8_t* __restrict__ lsb, float* __restrict__ scale,
                                                               int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh,
                                                               int B, int H, int cap, const int32_t* __restrict__ step) {
  constexpr int HALF = D / 2, LPR = D / 16, RPB = 64 / LPR;
  const int lane = threadIdx.x, c = lane % LPR;
  const int unit = blockIdx.x * RPB + lane / LPR;
  const int row = step[0] - 1;
  if (unit >= B * H || row < 0 || row >= cap) return;      // (whole LPR-groups leave together)
  const int b = unit / H, h = unit % H;
  using V8 = Vec8<T>;
  const T* rows = reinterpret_cast<const T*>(reinterpret_cast<const char*>(step) + kStepHeader);   // cos[2][HALF] | sin[2][HALF]
  const T* kp = k_new + b * new_sb + h * new_sh;
  const T* vp = v_new + b * new_sb + h * new_sh;
  const typename V8::raw k0 = V8::ldg(kp + 8 * c), k1 = V8::ldg(kp + HALF + 8 * c);
  const typename V8::raw v0 = V8::ldg(vp + 8 * c), v1 = V8::ldg(vp + HALF + 8 * c);
  float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
  V8::unpack(k0, xlo);
  V8::unpack(k1, xhi);
  V8::unpack(V8::ldg(rows + HALF + 8 * c), cc);
  V8::unpack(V8::ldg(rows + 3 * HALF + 8 * c), ss);
  rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
  const int64_t dst = b * kv_sb + h * kv_sh + (int64_t)row * D;
  if (kc) { V8::stg(kc + dst + 8 * c, k0); V8::stg(kc + dst + HALF + 8 * c, k1); }
  V8::stg(krc + dst + 8 * c, V8::pack(ylo));
  V8::stg(krc + dst + HALF + 8 * c, V8::pack(yhi));
  V8::stg(vc + dst + 8 * c, v0);
  V8::stg(vc + dst + HALF + 8 * c, v1);
  if (msb == nullptr) return;
  // the planes of the row (pq_pack_kernel's arithmetic; ylo / yhi ARE the values the shadow now holds: rope_pair rounds)
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fmaxf(fabsf(ylo[e]), fabsf(yhi[e])));
  amax = fmaxf(amax, dpp_mov<kDppXor1>(amax));
  amax = fmaxf(amax, dpp_mov<kDppXor2>(amax));
  if (LPR == 8) amax = fmaxf(amax, dpp_mov<kDppHalfMirror>(amax));
  const float sc = amax > 0.f ? amax / 127.0f : 1.0f;
  uint32_t m_lo = 0, l_lo = 0, m_hi = 0, l_hi = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int qa = (int)rintf(ylo[e] / sc), qb = (int)rintf(yhi[e] / sc);
    qa = max(-128, min(127, qa));
    qb = max(-128, min(127, qb));
    m_lo |= (uint32_t)((qa >> 4) & 15) << (4 * e);
    l_lo |= (uint32_t)(qa & 15) << (4 * e);
    m_hi |= (uint32_t)((qb >> 4) & 15) << (4 * e);
    l_hi |= (uint32_t)(qb & 15) << (4 * e);
  }
  const int64_t po = b * pl_sb + h * pl_sh + (int64_t)row * (D / 2);
  *reinterpret_cast<uint32_t*>(msb + po + 4 * c) = m_lo;
  *reinterpret_cast<uint32_t*>(lsb + po + 4 * c) = l_lo;
  *reinterpret_cast<uint32_t*>(msb + po + 4 * (LPR + c)) = m_hi;
  *reinterpret_cast<uint32_t*>(lsb + po + 4 * (LPR + c)) = l_hi;
  if (c == 0) scale[b * sc_sb + h * sc_sh + row] = sc;
}

// ---- expand: planes -> integer-valued keys for the matrix cores (progressive-quant prefill) ----------------------------
// k_msb[row][e] = 16 * sext(msb nibble), k_full[row][e] = that + lsb nibble — exact in bf16 / f16 (|value| <= 128) —
// contiguous [B,Hkv,rows,D]; kscale[row] = scale / sqrt(D).  One lane per 8 elements.
template <typename T, int D>
__global__ __launch_bounds__(256) void pq_expand_kernel(const uint8_t* __restrict__ msb, const uint8_t* __restrict__ lsb,
                                                        const float* __restrict__ scale, int64_t pl_sb, int64_t pl_sh,
                                                        int64_t sc_sb, int64_t sc_sh, T* __restrict__ k_msb,
                                                        T* __restrict__ k_full, float* __restrict__ kscale, int Hkv, int rows) {
  constexpr int LPR = D / 8;
  constexpr int RPB = 256 / LPR;
  const int tid = threadIdx.x, c = tid % LPR, r = tid / LPR;
  const int row = blockIdx.x * RPB + r;
  const int hkv = blockIdx.y, b = blockIdx.z;
  if (row >= rows) return;
  const int64_t po = b * pl_sb + hkv * pl_sh + (int64_t)row * (D / 2) + 4 * c;
  const uint32_t m4 = *reinterpret_cast<const uint32_t*>(msb + po);
  const uint32_t l4 = *reinterpret_cast<const uint32_t*>(lsb + po);
  float a[8], f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int m = (int)((m4 >> (4 * e)) & 15u), l = (int)((l4 >> (4 * e)) & 15u);
    const int ms = (m ^ 8) - 8;                       // sign-extend the 4-bit field
    a[e] = (float)(ms * 16);
    f[e] = (float)(ms * 16 + l);
  }
  const int64_t o = ((int64_t)(b * Hkv + hkv) * rows + row) * D + 8 * c;
  Vec8<T>::stg(k_msb + o, Vec8<T>::pack(a));
  Vec8<T>::stg(k_full + o, Vec8<T>::pack(f));
  if (c == 0) kscale[(int64_t)(b * Hkv + hkv) * rows + row] = scale[b * sc_sb + hkv * sc_sh + row] / sqrtf((float)D);
}

int pq_expand(int dtype, const void* msb, const void* lsb, const float* scale, int64_t pl_sb, int64_t pl_sh, int64_t sc_sb,
              int64_t sc_sh, void* k_msb, void* k_full, float* kscale, int batch, int kv_heads, int head_dim, int rows,
              hipStream_t stream) {
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  const int rpb = 256 / (head_dim / 8);
  const dim3 grid((unsigned)ceil_div(rows, rpb), (unsigned)kv_heads, (unsigned)batch);
#define SPATTEN_EXPAND(T, DD)                                                                                          \
  hipLaunchKernelGGL((pq_expand_kernel<T, DD>), grid, dim3(256), 0, stream, (const uint8_t*)msb, (const uint8_t*)lsb,   \
                     scale, pl_sb, pl_sh, sc_sb, sc_sh, (T*)k_msb, (T*)k_full, kscale, kv_heads, rows)
  if (dtype == SPATTEN_BF16) { if (head_dim == 128) SPATTEN_EXPAND(bf16_t, 128); else SPATTEN_EXPAND(bf16_t, 64); }
  else if (dtype == SPATTEN_F16) { if (head_dim == 128) SPATTEN_EXPAND(f16_t, 128); else SPATTEN_EXPAND(f16_t, 64); }
  else return SPATTEN_ERR_UNSUPPORTED;
#undef SPATTEN_EXPAND
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}


// ---- profiled planes (ABI 4): key MSB plane of 4 / 6 / 8 bits + 4-bit LSB plane, value plane of 8 / 6 bits ----------------
// (layouts: include/spatten.h "Bit profiles and the quantised VALUE plane")

// (pack_fields / quantise_piece / store_plane_pieces: pq_pack.h — shared with the MSB pass that appends by itself, pq_decode.hip)

// LPR = D/16 lanes per row (the decode kernels' value mapping): lane c owns elements [8c, 8c+8) and [D/2+8c, D/2+8c+8)
template <typename T, int D, int KB, int VB>
__global__ __launch_bounds__(256) void pq_pack_planes_kernel(const T* __restrict__ kr, const T* __restrict__ v,
                                                             int64_t kv_sb, int64_t kv_sh, const PlanesDev pl, int lo, int hi,
                                                             const int32_t* __restrict__ step) {
  constexpr int LPR = D / 16, RPB = 256 / LPR, HALF = D / 2;
  const int tid = threadIdx.x, c = tid % LPR, r = tid / LPR;
  int row = lo + blockIdx.x * RPB + r;
  if (step != nullptr) {                  // device-length form: the step's row only
    if (r != 0) return;
    row = step[0] - 1;
  }
  if (row < 0 || row >= hi) return;       // (whole LPR-groups leave together)
  const int hkv = blockIdx.y, b = blockIdx.z;
  using V8 = Vec8<T>;
  const int64_t src = b * kv_sb + hkv * kv_sh + (int64_t)row * D;
  float kx[16], vx[16];
  {
    float a[8], bq[8];
    V8::unpack(V8::ldg(kr + src + 8 * c), a);
    V8::unpack(V8::ldg(kr + src + HALF + 8 * c), bq);
#pragma unroll
    for (int e = 0; e < 8; ++e) { kx[e] = a[e]; kx[8 + e] = bq[e]; }
    V8::unpack(V8::ldg(v + src + 8 * c), a);
    V8::unpack(V8::ldg(v + src + HALF + 8 * c), bq);
#pragma unroll
    for (int e = 0; e < 8; ++e) { vx[e] = a[e]; vx[8 + e] = bq[e]; }
  }
  store_plane_pieces<D, KB, VB>(pl, b, hkv, row, c, kx, vx);
}

// ---- one decode step's append AND its plane rows in ONE launch (round 5): row `row` (device-length form: state word 0 - 1) of
// k / v, its rotation into the shadow (modify_llama.py:95-104) and that row of the profiled planes — bit for bit what
// kv_append_kernel followed by pq_pack_planes_kernel leave there (the rotated values ARE model-dtype values: rope_pair rounds).
// The two one-row launches cost 4.8 + 3.5 us per layer-step of BASELINE configs[4] under the graph; this one launch replaces them.
template <typename T, int D, int KB, int VB>
__global__ __launch_bounds__(64) void kv_append_planes_kernel(const T* __restrict__ k_new, const T* __restrict__ v_new,
                                                              int64_t new_sb, int64_t new_sh, T* __restrict__ kc,
                                                              T* __restrict__ krc, T* __restrict__ vc, int64_t kv_sb,
                                                              int64_t kv_sh, const PlanesDev pl, const T* __restrict__ cos,
                                                              const T* __restrict__ sin, int table_rows, int B, int H,
                                                              int row_host, int cap, const int32_t* __restrict__ step) {
  constexpr int HALF = D / 2, LPR = D / 16, RPB = 64 / LPR;
  const int lane = threadIdx.x, c = lane % LPR;
  const int unit = blockIdx.x * RPB + lane / LPR;
  const int row = step ? step[0] - 1 : row_host;
  if (unit >= B * H || row < 0 || row >= cap) return;      // (whole LPR-groups leave together)
  const int b = unit / H, h = unit % H;
  using V8 = Vec8<T>;
  const T* cr;
  const T* sr;
  if (step) {                                             // the state's staged rotary row 1 = the appended key's slot
    const T* rows = reinterpret_cast<const T*>(reinterpret_cast<const char*>(step) + kStepHeader);   // cos[2][HALF] | sin[2][HALF]
    cr = rows + HALF; sr = rows + 3 * HALF;
  } else {
    const int ps = min(row, table_rows - 1);
    cr = cos + (int64_t)ps * HALF; sr = sin + (int64_t)ps * HALF;
  }
  const T* kp = k_new + b * new_sb + h * new_sh;
  const T* vp = v_new + b * new_sb + h * new_sh;
  const typename V8::raw k0 = V8::ldg(kp + 8 * c), k1 = V8::ldg(kp + HALF + 8 * c);
  const typename V8::raw v0 = V8::ldg(vp + 8 * c), v1 = V8::ldg(vp + HALF + 8 * c);
  float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
  V8::unpack(k0, xlo);
  V8::unpack(k1, xhi);
  V8::unpack(V8::ldg(cr + 8 * c), cc);
  V8::unpack(V8::ldg(sr + 8 * c), ss);
  rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
  const int64_t dst = b * kv_sb + h * kv_sh + (int64_t)row * D;
  if (kc) { V8::stg(kc + dst + 8 * c, k0); V8::stg(kc + dst + HALF + 8 * c, k1); }
  V8::stg(krc + dst + 8 * c, V8::pack(ylo));
  V8::stg(krc + dst + HALF + 8 * c, V8::pack(yhi));
  V8::stg(vc + dst + 8 * c, v0);
  V8::stg(vc + dst + HALF + 8 * c, v1);
  float kx[16], vx[16], va[8], vb[8];
  V8::unpack(v0, va);
  V8::unpack(v1, vb);
#pragma unroll
  for (int e = 0; e < 8; ++e) { kx[e] = ylo[e]; kx[8 + e] = yhi[e]; vx[e] = va[e]; vx[8 + e] = vb[e]; }
  store_plane_pieces<D, KB, VB>(pl, b, h, row, c, kx, vx);
}

bool planes_to_dev(const spatten_pq_planes_t* p, PlanesDev& d) {
  if (!p || p->struct_size != sizeof(spatten_pq_planes_t) || !p->key_msb || !p->key_lsb || !p->key_scale || !p->val_q ||
      !p->val_scale || !p->msb_logit)
    return false;
  d.km = (uint8_t*)p->key_msb; d.kl = (uint8_t*)p->key_lsb; d.ks = p->key_scale; d.vq = (uint8_t*)p->val_q;
  d.vs = p->val_scale; d.lg = p->msb_logit;
  d.km_sb = p->km_sb; d.km_sh = p->km_sh; d.kl_sb = p->kl_sb; d.kl_sh = p->kl_sh; d.vq_sb = p->vq_sb; d.vq_sh = p->vq_sh;
  d.sc_sb = p->sc_sb; d.sc_sh = p->sc_sh; d.lg_sb = p->lg_sb; d.lg_sh = p->lg_sh;
  return true;
}
bool pq_profile_supported(int kb, int vb) { return (kb == 4 && vb == 8) || (kb == 8 && vb == 8) || (kb == 6 && vb == 6); }

}  // namespace spatten

using namespace spatten;

extern "C" int spatten_pq_pack(int dtype, const void* kr_cache, int64_t kv_sb, int64_t kv_sh, void* msb, void* lsb,
                               float* scale, int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh, int batch,
                               int kv_heads, int head_dim, int row_lo, int row_hi, void* stream) {
  if (!kr_cache || !msb || !lsb || !scale || batch <= 0 || kv_heads <= 0 || row_lo < 0) return SPATTEN_ERR_INVALID;
  if (dtype != SPATTEN_F32 && dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  if (row_hi <= row_lo) return SPATTEN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 256 / (head_dim / 8);
  const dim3 grid((unsigned)ceil_div(row_hi - row_lo, rpb), (unsigned)kv_heads, (unsigned)batch);
#define SPATTEN_PACK(DD)                                                                                          \
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((pq_pack_kernel<T, DD>), grid, dim3(256), 0, st, (const T*)kr_cache, kv_sb, \
                                             kv_sh, (uint8_t*)msb, (uint8_t*)lsb, scale, pl_sb, pl_sh, sc_sb, sc_sh,     \
                                             row_lo, row_hi))
  if (head_dim == 128) { SPATTEN_PACK(128); } else { SPATTEN_PACK(64); }
#undef SPATTEN_PACK
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_kv_append_step(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh,
                                      void* k_cache, void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, void* msb,
                                      void* lsb, float* scale, int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh,
                                      int batch, int kv_heads, int head_dim, int capacity, const void* step_state,
                                      void* stream) {
  if (!k_new || !v_new || !kr_cache || !v_cache || !step_state || batch <= 0 || kv_heads <= 0 || capacity <= 0)
    return SPATTEN_ERR_INVALID;
  if ((msb == nullptr) != (lsb == nullptr) || (msb == nullptr) != (scale == nullptr)) return SPATTEN_ERR_INVALID;
  if (dtype != SPATTEN_F32 && dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 64 / (head_dim / 16);
  const dim3 grid((unsigned)ceil_div(batch * kv_heads, rpb));
#define SPATTEN_APPEND_STEP(DD)                                                                                          \
  SPATTEN_BY_DTYPE(dtype, hipLaunchKernelGGL((kv_append_pq_step_kernel<T, DD>), grid, dim3(64), 0, st, (const T*)k_new,     \
                                             (const T*)v_new, new_sb, new_sh, (T*)k_cache, (T*)kr_cache, (T*)v_cache, kv_sb, \
                                             kv_sh, (uint8_t*)msb, (uint8_t*)lsb, scale, pl_sb, pl_sh, sc_sb, sc_sh, batch,  \
                                             kv_heads, capacity, (const int32_t*)step_state))
  if (head_dim == 128) { SPATTEN_APPEND_STEP(128); } else { SPATTEN_APPEND_STEP(64); }
#undef SPATTEN_APPEND_STEP
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

// The decode step over the planes: spatten_attn_decode_args with the pq_* fields set (decode_attn.hip, KSRC = 1 / 2).

extern "C" size_t spatten_pq_plane_row_bytes(int head_dim, int bits) {
  if (head_dim <= 0 || bits <= 0 || (head_dim * bits) % 8 != 0) return 0;
  return (size_t)head_dim * bits / 8;
}

extern "C" int spatten_pq_pack_planes(int dtype, const void* kr_cache, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                                      const spatten_pq_planes_t* planes, int batch, int kv_heads, int head_dim, int row_lo,
                                      int row_hi, const void* step_state, void* stream) {
  PlanesDev pd;
  if (!kr_cache || !v_cache || !planes_to_dev(planes, pd) || batch <= 0 || kv_heads <= 0 || row_lo < 0) return SPATTEN_ERR_INVALID;
  if (!ok_dtype(dtype)) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  const int kb = planes->key_msb_bits, vb = planes->value_bits;
  if (!pq_profile_supported(kb, vb)) return SPATTEN_ERR_UNSUPPORTED;
  if (!step_state && row_hi <= row_lo) return SPATTEN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 256 / (head_dim / 16);
  const dim3 grid((unsigned)(step_state ? 1 : ceil_div(row_hi - row_lo, rpb)), (unsigned)kv_heads, (unsigned)batch);
#define SPATTEN_PACKP(T, DD, KB, VB)                                                                                        \
  hipLaunchKernelGGL((pq_pack_planes_kernel<T, DD, KB, VB>), grid, dim3(256), 0, st, (const T*)kr_cache, (const T*)v_cache,  \
                     kv_sb, kv_sh, pd, row_lo, row_hi, (const int32_t*)step_state)
#define SPATTEN_PACKP_D(T, KB, VB) do { if (head_dim == 128) SPATTEN_PACKP(T, 128, KB, VB); else SPATTEN_PACKP(T, 64, KB, VB); } while (0)
#define SPATTEN_PACKP_T(KB, VB) do { if (dtype == SPATTEN_BF16) SPATTEN_PACKP_D(bf16_t, KB, VB); else if (dtype == SPATTEN_F16) SPATTEN_PACKP_D(f16_t, KB, VB); else SPATTEN_PACKP_D(float, KB, VB); } while (0)
  if (kb == 4) SPATTEN_PACKP_T(4, 8);
  else if (kb == 8) SPATTEN_PACKP_T(8, 8);
  else SPATTEN_PACKP_T(6, 6);
#undef SPATTEN_PACKP_T
#undef SPATTEN_PACKP_D
#undef SPATTEN_PACKP
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_kv_append_planes(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh,
                                        void* k_cache, void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh,
                                        const spatten_pq_planes_t* planes, const void* cos, const void* sin, int table_rows,
                                        int batch, int kv_heads, int head_dim, int row, int capacity, const void* step_state,
                                        void* stream) {
  PlanesDev pd;
  if (!k_new || !v_new || !kr_cache || !v_cache || !planes_to_dev(planes, pd) || batch <= 0 || kv_heads <= 0 || capacity <= 0)
    return SPATTEN_ERR_INVALID;
  if (!step_state && (!cos || !sin || table_rows <= 0 || row < 0 || row >= capacity)) return SPATTEN_ERR_INVALID;
  if (!ok_dtype(dtype)) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  const int kb = planes->key_msb_bits, vb = planes->value_bits;
  if (!pq_profile_supported(kb, vb)) return SPATTEN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 64 / (head_dim / 16);
  const dim3 grid((unsigned)ceil_div(batch * kv_heads, rpb));
#define SPATTEN_APL(T, DD, KB, VB)                                                                                            \
  hipLaunchKernelGGL((kv_append_planes_kernel<T, DD, KB, VB>), grid, dim3(64), 0, st, (const T*)k_new, (const T*)v_new, new_sb, \
                     new_sh, (T*)k_cache, (T*)kr_cache, (T*)v_cache, kv_sb, kv_sh, pd, (const T*)cos, (const T*)sin, table_rows, \
                     batch, kv_heads, row, capacity, (const int32_t*)step_state)
#define SPATTEN_APL_D(T, KB, VB) do { if (head_dim == 128) SPATTEN_APL(T, 128, KB, VB); else SPATTEN_APL(T, 64, KB, VB); } while (0)
#define SPATTEN_APL_T(KB, VB) do { if (dtype == SPATTEN_BF16) SPATTEN_APL_D(bf16_t, KB, VB); else if (dtype == SPATTEN_F16) SPATTEN_APL_D(f16_t, KB, VB); else SPATTEN_APL_D(float, KB, VB); } while (0)
  if (kb == 4) SPATTEN_APL_T(4, 8);
  else if (kb == 8) SPATTEN_APL_T(8, 8);
  else SPATTEN_APL_T(6, 6);
#undef SPATTEN_APL_T
#undef SPATTEN_APL_D
#undef SPATTEN_APL
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}
