// common.h — device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/spatten.h"

namespace spatten {

constexpr int kWave = 64;  // gfx950 wavefront

using bf16_t = __bf16;
using f16_t = _Float16;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Model-dtype traits.  `round(x)` = the value torch stores when an fp32 opmath result is written to a
// tensor of the model dtype (round-to-nearest-even), returned as fp32.
// ------------------------------------------------------------------------------------------------
template <typename T> struct DT;

template <> struct DT<float> {
  static constexpr int kId = SPATTEN_F32;
  static constexpr bool k16 = false;
  __device__ static inline float to_f32(float x) { return x; }
  __device__ static inline float from_f32(float x) { return x; }
  __device__ static inline float round(float x) { return x; }
};

template <> struct DT<f16_t> {
  static constexpr int kId = SPATTEN_F16;
  static constexpr bool k16 = true;
  __device__ static inline float to_f32(f16_t x) { return (float)x; }
  __device__ static inline f16_t from_f32(float x) { return (f16_t)x; }
  __device__ static inline float round(float x) { return (float)(f16_t)x; }
};

template <> struct DT<bf16_t> {
  static constexpr int kId = SPATTEN_BF16;
  static constexpr bool k16 = true;
  __device__ static inline float to_f32(bf16_t x) { return (float)x; }
  __device__ static inline bf16_t from_f32(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32 (RNE)
  // ONE instruction: v_cvt_pk_bf16_f32 d, 0, x puts bf16(x) in the upper half of d over a zero lower half — which IS the
  // fp32 encoding of the rounded value (r05; the cast pair (float)(bf16_t)x costs a convert and a shift, the packed
  // convert of two values a convert, a shift and a mask)
  __device__ static inline float round(float x) {
#if defined(SPATTEN_ROUND_VIA_CAST) && SPATTEN_ROUND_VIA_CAST   // A/B only (tools/mb/pf_exp.sh): the two-instruction form of r01-r04
    return (float)(bf16_t)x;
#endif
    typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
    bf2 b = __builtin_convertvector(f32x2{0.f, x}, bf2);
    return __uint_as_float(*reinterpret_cast<uint32_t*>(&b));
  }
};

// round(x) on a pair (kept for its callers; one instruction per value for bf16 since r05, see DT<bf16_t>::round)
template <typename T> __device__ inline f32x2 round2(f32x2 v) { return f32x2{DT<T>::round(v[0]), DT<T>::round(v[1])}; }
// div_by_const (below) on a pair: packed fp32 multiply / fma (v_pk_mul_f32, v_pk_fma_f32)
__device__ inline f32x2 div_by_const2(f32x2 x, float c, float rc) {
  const f32x2 c2 = {c, c}, rc2 = {rc, rc};
  const f32x2 y = x * rc2;
  const f32x2 e = __builtin_elementwise_fma(-y, c2, x);
  return __builtin_elementwise_fma(e, rc2, y);
}

// 8 consecutive elements of the model dtype, kept packed ("raw") until used.  16-bit types: one
// 16-byte access; fp32: two.  Pointers must be 16-byte aligned (head_dim % 8 == 0, rows pitch d).
template <typename T> struct Vec8;

template <> struct Vec8<float> {
  struct raw { f32x4 a, b; };
  __device__ static inline raw ldg(const float* p) {
    raw r;
    r.a = *reinterpret_cast<const f32x4*>(p);
    r.b = *reinterpret_cast<const f32x4*>(p + 4);
    return r;
  }
  __device__ static inline raw ldg_stream(const float* p) {
    raw r;
    r.a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    r.b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    return r;
  }
  __device__ static inline void stg(float* p, const raw& r) {
    *reinterpret_cast<f32x4*>(p) = r.a;
    *reinterpret_cast<f32x4*>(p + 4) = r.b;
  }
  __device__ static inline void unpack(const raw& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
  }
  __device__ static inline raw pack(const float (&v)[8]) {
    raw r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.a[i] = v[i]; r.b[i] = v[4 + i]; }
    return r;
  }
};

template <> struct Vec8<bf16_t> {
  using raw = u32x4;
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  __device__ static inline raw ldg(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
  __device__ static inline raw ldg_stream(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  __device__ static inline void stg(bf16_t* p, const raw& r) { *reinterpret_cast<u32x4*>(p) = r; }
  __device__ static inline void unpack(const raw& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(r[i] << 16);
      v[2 * i + 1] = __uint_as_float(r[i] & 0xFFFF0000u);
    }
  }
  __device__ static inline raw pack(const float (&v)[8]) {
    raw r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 f = {v[2 * i], v[2 * i + 1]};
      bf2 b = __builtin_convertvector(f, bf2);
      r[i] = *reinterpret_cast<uint32_t*>(&b);
    }
    return r;
  }
};

template <> struct Vec8<f16_t> {
  using raw = u32x4;
  typedef f16_t h2 __attribute__((ext_vector_type(2)));
  __device__ static inline raw ldg(const f16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
  __device__ static inline raw ldg_stream(const f16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  __device__ static inline void stg(f16_t* p, const raw& r) { *reinterpret_cast<u32x4*>(p) = r; }
  __device__ static inline void unpack(const raw& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t w = r[i];
      h2 h = *reinterpret_cast<h2*>(&w);
      v[2 * i] = (float)h[0];
      v[2 * i + 1] = (float)h[1];
    }
  }
  __device__ static inline raw pack(const float (&v)[8]) {
    raw r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h2 h = {(f16_t)v[2 * i], (f16_t)v[2 * i + 1]};
      r[i] = *reinterpret_cast<uint32_t*>(&h);
    }
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// wave64 cross-lane helpers.  DPP / permlane-swap forms run at VALU rate (a ds_bpermute shuffle is an
// LDS round trip of ~100+ cycles, which is what a latency-bound decode kernel cannot afford).
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ inline float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7-i within 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15-i within 16
constexpr int kDppRor4 = 0x124;        // rotate right by 4 within 16
constexpr int kDppRor8 = 0x128;        // rotate right by 8 within 16 (== xor 8)

// identity the optimiser cannot see through: keeps a cheap per-lane computation inside a loop instead of hoisted
// (and then spilled) above it
__device__ inline int opaque_lane(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// max of three without the canonicalising v_max the compiler puts in front of fmaxf on values it cannot prove quiet
// (MFMA results, bit-built floats): one VALU issue for two new values.  NaN inputs are not expected here.
__device__ inline float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// sum over aligned groups of WIDTH lanes (4, 8 or 16); every lane of the group gets the total
template <int WIDTH>
__device__ inline float group_sum(float v) {
  static_assert(WIDTH == 4 || WIDTH == 8 || WIDTH == 16, "group width");
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  if (WIDTH >= 8) v += dpp_mov<kDppHalfMirror>(v);
  if (WIDTH >= 16) v += dpp_mov<kDppMirror>(v);
  return v;
}
// x[i] + x[i^16] and x[i] + x[i^32] in every lane (gfx950 v_permlane16_swap / v_permlane32_swap)
__device__ inline float xor16_sum(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float xor32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float wave_max(float v) {
  v = fmaxf(v, dpp_mov<kDppXor1>(v));
  v = fmaxf(v, dpp_mov<kDppXor2>(v));
  v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
  v = fmaxf(v, dpp_mov<kDppMirror>(v));
  return xor32_max(xor16_max(v));
}
__device__ inline float wave_sum(float v) { return xor32_sum(xor16_sum(group_sum<16>(v))); }

// acc + both halves of a packed model-dtype pair, on the packed-dot unit (v_dot2c_f32_bf16 / v_dot2_f32_f16 against (1, 1))
template <typename T> __device__ inline float pair_sum(uint32_t packed, float acc);
template <> __device__ inline float pair_sum<bf16_t>(uint32_t packed, float acc) {
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  const uint32_t ones = 0x3F803F80u;
  return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2*>(&packed), *reinterpret_cast<const bf2*>(&ones), acc, false);
}
template <> __device__ inline float pair_sum<f16_t>(uint32_t packed, float acc) {
  typedef f16_t h2 __attribute__((ext_vector_type(2)));
  const uint32_t ones = 0x3C003C00u;
  return __builtin_amdgcn_fdot2(*reinterpret_cast<h2*>(&packed), *reinterpret_cast<const h2*>(&ones), acc, false);
}
template <> __device__ inline float pair_sum<float>(uint32_t packed, float acc) { return acc + __uint_as_float(packed); }

// dot product of 8 packed model-dtype pairs with fp32 accumulation (v_dot2c_f32_bf16 / v_dot2_f32_f16)
template <typename T> struct Dot8;
template <> struct Dot8<float> {
  using packed = Vec8<float>::raw;
  __device__ static inline packed pack(const float (&v)[8]) { return Vec8<float>::pack(v); }
  __device__ static inline float dot(const packed& a, const packed& b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc = fmaf(a.a[i], b.a[i], acc); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc = fmaf(a.b[i], b.b[i], acc); }
    return acc;
  }
};
template <> struct Dot8<bf16_t> {
  using packed = u32x4;
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  __device__ static inline packed pack(const float (&v)[8]) { return Vec8<bf16_t>::pack(v); }
  __device__ static inline float dot(const packed& a, const packed& b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t x = a[i], y = b[i];
      acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2*>(&x), *reinterpret_cast<bf2*>(&y), acc, false);
    }
    return acc;
  }
};
template <> struct Dot8<f16_t> {
  using packed = u32x4;
  typedef f16_t h2 __attribute__((ext_vector_type(2)));
  __device__ static inline packed pack(const float (&v)[8]) { return Vec8<f16_t>::pack(v); }
  __device__ static inline float dot(const packed& a, const packed& b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t x = a[i], y = b[i];
      acc = __builtin_amdgcn_fdot2(*reinterpret_cast<h2*>(&x), *reinterpret_cast<h2*>(&y), acc, false);
    }
    return acc;
  }
};

// o[e] += p0 * x0[e] + p1 * x1[e] over two rows of 8 model-dtype elements (P·V of the decode kernel, two keys at a
// time).  16-bit types: v_perm_b32 pairs element e of the two rows into one dword and the packed dot instruction takes
// the probability pair (rounded to the model dtype — the reference rounds P to the dtype before P·V as well,
// modify_llama.py:135-138) against it: 2 VALU ops per 2 products instead of 4 (unpack, unpack, fma, fma).
template <typename T> struct PairFma;
template <> struct PairFma<float> {
  using prob2 = f32x2;
  __device__ static inline prob2 pack_p(float p0, float p1) { return f32x2{p0, p1}; }
  __device__ static inline void fma(float (&o)[8], const Vec8<float>::raw& x0, const Vec8<float>::raw& x1, prob2 pp) {
    float a[8], b[8];
    Vec8<float>::unpack(x0, a);
    Vec8<float>::unpack(x1, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(pp[1], b[e], fmaf(pp[0], a[e], o[e]));
  }
};
template <> struct PairFma<bf16_t> {
  using prob2 = uint32_t;
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  __device__ static inline prob2 pack_p(float p0, float p1) {
    bf2 b = __builtin_convertvector(f32x2{p0, p1}, bf2);
    return *reinterpret_cast<uint32_t*>(&b);
  }
  __device__ static inline void fma(float (&o)[8], const u32x4& x0, const u32x4& x1, prob2 pp) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t lo = __builtin_amdgcn_perm(x1[i], x0[i], 0x05040100u), hi = __builtin_amdgcn_perm(x1[i], x0[i], 0x07060302u);
      o[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2*>(&lo), *reinterpret_cast<bf2*>(&pp), o[2 * i], false);
      o[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2*>(&hi), *reinterpret_cast<bf2*>(&pp), o[2 * i + 1], false);
    }
  }
};
template <> struct PairFma<f16_t> {
  using prob2 = uint32_t;
  typedef f16_t h2 __attribute__((ext_vector_type(2)));
  __device__ static inline prob2 pack_p(float p0, float p1) {
    h2 h = {(f16_t)p0, (f16_t)p1};
    return *reinterpret_cast<uint32_t*>(&h);
  }
  __device__ static inline void fma(float (&o)[8], const u32x4& x0, const u32x4& x1, prob2 pp) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t lo = __builtin_amdgcn_perm(x1[i], x0[i], 0x05040100u), hi = __builtin_amdgcn_perm(x1[i], x0[i], 0x07060302u);
      o[2 * i] = __builtin_amdgcn_fdot2(*reinterpret_cast<h2*>(&lo), *reinterpret_cast<h2*>(&pp), o[2 * i], false);
      o[2 * i + 1] = __builtin_amdgcn_fdot2(*reinterpret_cast<h2*>(&hi), *reinterpret_cast<h2*>(&pp), o[2 * i + 1], false);
    }
  }
};

// dot product of 8 query elements with the 8 unsigned nibbles of one dword of a 4-bit plane (progressive-quant keys).
// 16-bit types: OR a nibble pair into the mantissas of two constants (bf16 128.0 / f16 1024.0 -> exactly 128+n /
// 1024+n), feed the pair to the packed dot instruction against the query pair, and take the constant out again
// through the precomputed query sum: 3 VALU ops per 2 elements instead of ~5 per element for extract/convert/fma.
// Signed (MSB) nibbles are XOR-ed with 8 first (n ^ 8 = sext(n) + 8), which only changes the constant.
// Returns sum_e q[e] * n_e with n_e = nibble e of w, minus `bias` * sum(q) (bias = 8 turns n ^ 8 back into sext(n)).
template <typename T> struct NibbleDot;
template <> struct NibbleDot<float> {
  struct packed { float q[8]; float qsum; };
  __device__ static inline packed prep(const float (&v)[8]) {
    packed r;
    r.qsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.q[i] = v[i]; r.qsum += v[i]; }
    return r;
  }
  __device__ static inline float dot(const packed& a, uint32_t w, float bias) {
    float acc = -bias * a.qsum;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(a.q[e], (float)((w >> (4 * e)) & 15u), acc);
    return acc;
  }
};
// (x & mask) | magic as ONE v_and_or_b32 (round 5): left to the compiler, two literals become v_and_b32 + v_or_b32 — a VOP3 cannot
// carry a literal on gfx950, so the mask rides in an SGPR and the magic in a VGPR, both loop invariant
__device__ inline uint32_t and_or_b32(uint32_t x, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
  return r;
}
template <typename T, uint32_t MAGIC, int BASE>
struct NibbleDot16 {
  struct packed { uint32_t q[4]; float qsum; };   // q[k] = {element k, element k + 4}
  __device__ static inline packed prep(const float (&v)[8]) {
    packed r;
    const float w[8] = {v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7]};
    const u32x4 pk = Vec8<T>::pack(w);
    r.qsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.q[i] = pk[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r.qsum += v[i];   // v is already rounded to T: the packed values are exact
    return r;
  }
  __device__ static inline float dot(const packed& a, uint32_t w, float bias) {
    u32x4 x, y;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = a.q[k]; y[k] = and_or_b32(w >> (4 * k), 0x000F000Fu, MAGIC); }
    return Dot8<T>::dot(x, y, -((float)BASE + bias) * a.qsum);
  }
};
template <> struct NibbleDot<bf16_t> : NibbleDot16<bf16_t, 0x43004300u, 128> {};
template <> struct NibbleDot<f16_t> : NibbleDot16<f16_t, 0x64006400u, 1024> {};

// Monotone fp32 -> uint32 key: larger value => larger key, NaN largest, -0 == +0.
// This is the total order torch.topk(largest=True) ranks by.
__device__ inline uint32_t ordered_key(float x) {
  if (x != x) return 0xFFFFFFFFu;
  if (x == 0.0f) x = 0.0f;  // canonicalise -0
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// x / c for the logit scale (modify_llama.py:111-113) as one Newton correction on the reciprocal product: 3 VALU
// ops instead of the ~15 of the IEEE divide sequence.  x is a model-dtype value and the quotient is rounded to the
// model dtype right after, so a last-bit difference of the fp32 quotient can only matter exactly on a rounding
// boundary of the 16-bit type (and for fp32 it is within the stated 1e-5 tolerance).  d = 64 / 256: exact.
__device__ inline float div_by_const(float x, float c, float rc) {
  const float y = x * rc;
  const float e = fmaf(-y, c, x);
  return fmaf(e, rc, y);
}

// The reference's "logits / sqrt(d) -> dtype" step (modify_llama.py:111-113) for a logit x that already IS a model-dtype
// value.  bf16: round(x * (1/c)) equals round(x / c) for EVERY bf16 x at c = sqrt(128) (all 128 mantissas x all
// exponents enumerated against the fp32 division; c = 8 and 16 are exact anyway), so the Newton correction is not
// needed; f16 (11-bit mantissa) has inputs where it is.
template <typename T> __device__ inline float logit_scale(float x, float c, float rc) { return div_by_const(x, c, rc); }
template <> __device__ inline float logit_scale<bf16_t>(float x, float c, float rc) { return x * rc; }

template <typename T>
__device__ inline void rope_pair(const float (&xlo)[8], const float (&xhi)[8], const float (&c)[8],
                                 const float (&s)[8], float (&ylo)[8], float (&yhi)[8]) {
  // y = x*cos + rotate_half(x)*sin with rotate_half(x) = cat(-x[d/2:], x[:d/2])   (modify_llama.py:21-28)
  // each of the three torch ops rounds to the model dtype.
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a_lo = DT<T>::round(xlo[i] * c[i]);
    const float b_lo = DT<T>::round(-xhi[i] * s[i]);
    const float a_hi = DT<T>::round(xhi[i] * c[i]);
    const float b_hi = DT<T>::round(xlo[i] * s[i]);
    ylo[i] = DT<T>::round(a_lo + b_lo);
    yhi[i] = DT<T>::round(a_hi + b_hi);
  }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Per-device facts the host side sizes co-resident grids by: asked once per device, not assumed.  (Dynamic LDS
// requests are checked through hipFuncSetAttribute's own answer, per device: local_v.hip, layer_cascade.hip.)
struct DeviceFacts { int cus = 0; int lds_per_block = 0; };
inline const DeviceFacts& device_facts() {
  static DeviceFacts facts[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  DeviceFacts& f = facts[dev];
  if (f.cus == 0) {
    int c = 0, l = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    if (hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || l <= 0) l = 64 * 1024;
    f.lds_per_block = l;
    f.cus = c;
  }
  return f;
}
// Workgroups of a launch that the device starts without waiting for another to finish: one per CU — what a merge that POLLS
// for its sibling splits needs (every workgroup of the grid resident).  A larger grid takes the ticket protocol.  The bound
// assumes the launch has the device to itself: work on other streams (RCCL, a second model) can delay residency — the spins
// are bounded and a timeout is reported through the workspace status (spatten_decode_workspace_status).
inline int coresident_workgroups() { return device_facts().cus; }

// progressive-quant key planes handed to decode_rows (decode_attn.hip); see pq.hip for the storage format
struct PQKeys {
  const uint8_t* msb; const uint8_t* lsb; const float* scale;
  int64_t pl_sb, pl_sh, sc_sb, sc_sh;
  float threshold; int32_t* need;
};

// Host-side description of one launch of the decode kernel family (decode_attn.hip); filled by the C-ABI entry points
// (spatten_attn_decode*, the rows leg of spatten_attn_prefill).  Strides in elements.
struct DecodeCall {
  int dtype = 0;
  const void* q = nullptr; int64_t q_sb = 0, q_sh = 0, q_sq = 0;
  void* k_cache = nullptr; void* kr_cache = nullptr; void* v_cache = nullptr; int64_t kv_sb = 0, kv_sh = 0;
  const void* k_new = nullptr; const void* v_new = nullptr; int64_t new_sb = 0, new_sh = 0;
  const void* cos = nullptr; const void* sin = nullptr; int table_rows = 0;
  const int64_t* position_ids = nullptr; int64_t pos_sb = 0;
  const void* mask = nullptr; int64_t mask_sb = 0, mask_sq = 0;
  void* out = nullptr; int64_t out_sb = 0, out_sq = 0;
  void* scores = nullptr; int64_t sc_sb = 0, sc_sh = 0, sc_sq = 0;
  float* lse = nullptr; int lse_q = 0;   // lse rows per (b, h) (0 = n_q)
  // split-N workspace: [256 B header: word 0 = error flag][units x {counter, generation}][units x ws_splits x (D+2) granules]
  void* workspace = nullptr; size_t ws_units = 0; int ws_splits = 0;
  int batch = 0, heads = 0, kv_heads = 0, head_dim = 0, kv_len = 0, pos_q = 0, n_q = 1, causal = 0, n_splits = 0;
  int vis0 = 0;   // causal: keys visible to query row 0 (0 = kv_len - n_q + 1, the HF rule for a block that ENDS the cache)
  const int32_t* head_ids = nullptr; int n_active = 0; int flags = 0;
  const PQKeys* pq = nullptr;
  // cascade importance, deferred by one step: the PREVIOUS step's stash + (max, sum) are folded into acc by this launch
  const void* prev_scores = nullptr; int64_t pv_sb = 0, pv_sh = 0; const float* prev_lse = nullptr;
  float* acc = nullptr; int64_t acc_sh = 0; int prev_len = 0;
  float* head_abs = nullptr;   // [B*H] head importance accumulators (n_q == 1 only)
  const void* step = nullptr;  // device-resident step state (step.hip): kv_len above is then a bound
  int layout_len = 0;          // > kv_len: lay the splits out for this length (spatten_decode_args_t::kv_len_layout)
  // fused / trailing output projection (spatten_decode_args_t::proj_*)
  const void* proj_w = nullptr; int64_t proj_w_sn = 0; const void* proj_bias = nullptr; void* proj_out = nullptr;
  int64_t proj_out_sb = 0; int proj_n = 0;
  // the step's q / k / v projections fused into the launch (spatten_decode_args_t::qkv_*)
  const void* qkv_x = nullptr; const void* qkv_w = nullptr; int64_t qkv_w_sn = 0; const void* qkv_bias = nullptr;
  void* qkv_xch = nullptr; int qkv_hidden = 0;
  bool step_oproj_off = false;   // (reserved: keep the output projection a separate launch)
};
int decode_rows(const DecodeCall& c, hipStream_t stream);
// the grouped-query single-row step on the matrix cores (decode_gqa.hip): SPATTEN_OK after launching it, SPATTEN_ERR_UNSUPPORTED when
// the launch is not one it serves (nothing launched; decode_rows then takes the per-query-head kernel)
int decode_gqa_rows(const DecodeCall& c, hipStream_t stream);
// split-N factor of a decode launch over `units` softmax rows (decode_attn.hip); dense_rule = the 16-bit dense step's 8-split rule
int decode_auto_splits(int units, int d, int kv_len, int elt, bool dense_rule);
int decode_team();   // threads of the attention team of a single-shot decode step (decode_attn.hip: 512 or 256)
// y[m, n] = sum_k x[m, k] W[n, k] (+ bias): the weight-streaming kernel of gemv.hip (C++ linkage for the other units)
int gemv_rows(int dtype, const void* x, int64_t x_sm, const void* W, int64_t w_sn, const void* bias, void* y, int64_t y_sm,
              int M, int N, int K, hipStream_t stream);

// planes -> integer-valued keys in the model dtype (msb*16 and msb*16+lsb, both exact) + scale / sqrt(d) per key
// (pq.hip; used by the progressive-quant prefill)
int pq_expand(int dtype, const void* msb, const void* lsb, const float* scale, int64_t pl_sb, int64_t pl_sh, int64_t sc_sb,
              int64_t sc_sh, void* k_msb, void* k_full, float* kscale, int batch, int kv_heads, int head_dim, int rows,
              hipStream_t stream);

// profiled planes (pq.hip / pq_decode.hip): the device-side copy of spatten_pq_planes_t
struct PlanesDev {
  uint8_t* km; uint8_t* kl; float* ks; uint8_t* vq; float* vs; float* lg;
  int64_t km_sb, km_sh, kl_sb, kl_sh, vq_sb, vq_sh, sc_sb, sc_sh, lg_sb, lg_sh;
};
bool planes_to_dev(const spatten_pq_planes_t* p, PlanesDev& d);
bool pq_profile_supported(int key_msb_bits, int value_bits);

constexpr int kDecodeMaxSplits = 64;
constexpr size_t kStepHeader = 64;      // spatten_step_state_t: 16 int32 words, then the staged rotary rows
constexpr size_t kDecodeWsHeader = 256;
inline size_t decode_cnt_bytes(size_t units) { return (units * 2 * sizeof(unsigned) + 255) / 256 * 256; }

static inline bool ok_dtype(int dt) { return dt == SPATTEN_F32 || dt == SPATTEN_F16 || dt == SPATTEN_BF16; }

}  // namespace spatten

// CALL with `T` bound to the element type of the dtype enum (inside namespace spatten or after `using namespace`)
#define SPATTEN_BY_DTYPE(dt, CALL)                                 \
  switch (dt) {                                                    \
    case SPATTEN_F32: { using T = float; CALL; } break;            \
    case SPATTEN_F16: { using T = spatten::f16_t; CALL; } break;   \
    default: { using T = spatten::bf16_t; CALL; }                  \
  }
