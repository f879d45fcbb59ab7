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
  __device__ static inline float round(float x) { return (float)(bf16_t)x; }
};

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

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace spatten
