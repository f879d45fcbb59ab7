/*
 * oracle.c — plain-C restatement of the reference's algorithm for the pruned-attention hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Built by `make oracle` into oracle/liboracle.so and used only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ("port" of the reference timed on
 * the GPU box's host cores).  It does what the reference does, op for op — including re-rotating the
 * whole un-rotated K cache on every decode step — with the rounding to the model dtype after every torch
 * op.  Pinned against the golden vectors captured from the imported reference (tests/test_c_oracle.py).
 *
 * Reference lines restated (paths relative to mit-han-lab/spatten):
 *   orc_attn_decode   spatten_llm/pos_shift/modify_llama.py:86-147 at q_len == 1
 *   orc_topk_window   spatten_llm/kv_cache_token_pruning.py:59-63
 *   orc_kv_compact    spatten_llm/kv_cache_token_pruning.py:64-96
 *
 * Storage: dtype 0 = float32, 1 = float16 (uint16 bits), 2 = bfloat16 (uint16 bits).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_BF16 = 2 };

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
  if (exp == 0) {
    if (man == 0) return u2f(sign);
    int e = -1;
    do { man <<= 1; ++e; } while (!(man & 0x400u));
    return u2f(sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13));
  }
  if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
  return u2f(sign | ((exp + 112) << 23) | (man << 13));
}
static inline uint16_t f32_to_f16(float f) {   /* round to nearest even */
  const uint32_t u = f2u(f), sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);          /* >= 65536 -> inf (65520 rounds to inf below) */
  if (a < 0x38800000u) {                                            /* subnormal half or zero */
    if (a < 0x33000000u) return (uint16_t)sign;
    const int shift = 113 - (int)(a >> 23);
    uint32_t man = (a & 0x7fffffu) | 0x800000u;
    const uint32_t lost = man & ((1u << (shift + 13)) - 1u), half = 1u << (shift + 12);
    man >>= (shift + 13);
    if (lost > half || (lost == half && (man & 1u))) ++man;
    return (uint16_t)(sign | man);
  }
  uint32_t r = a - 0x38000000u;                                      /* rebias */
  const uint32_t lost = r & 0x1fffu;
  r >>= 13;
  if (lost > 0x1000u || (lost == 0x1000u && (r & 1u))) ++r;          /* may carry into inf: correct */
  return (uint16_t)(sign | r);
}

static inline float rnd(float x, int dt) {
  if (dt == ORC_BF16) return bf16_to_f32(f32_to_bf16(x));
  if (dt == ORC_F16) return f16_to_f32(f32_to_f16(x));
  return x;
}
static inline float ld(const void* p, int64_t i, int dt) {
  if (dt == ORC_F32) return ((const float*)p)[i];
  if (dt == ORC_F16) return f16_to_f32(((const uint16_t*)p)[i]);
  return bf16_to_f32(((const uint16_t*)p)[i]);
}
static inline void st(void* p, int64_t i, float v, int dt) {
  if (dt == ORC_F32) ((float*)p)[i] = v;
  else if (dt == ORC_F16) ((uint16_t*)p)[i] = f32_to_f16(v);
  else ((uint16_t*)p)[i] = f32_to_bf16(v);
}
static inline int esz(int dt) { return dt == ORC_F32 ? 4 : 2; }

void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_max_threads(void) { return omp_get_max_threads(); }

/* apply_rotary_pos_emb_single on one row (modify_llama.py:21-28): three rounded ops.
 * cos/sin: half tables [rows, d/2] (emb = cat(freqs, freqs)). */
static void rope_row(const float* x, float* y, const void* cos, const void* sin, int64_t pos, int d, int dt) {
  const int h = d / 2;
  for (int i = 0; i < h; ++i) {
    const float c = ld(cos, pos * h + i, dt), s = ld(sin, pos * h + i, dt);
    const float a_lo = rnd(x[i] * c, dt), b_lo = rnd(-x[i + h] * s, dt);
    const float a_hi = rnd(x[i + h] * c, dt), b_hi = rnd(x[i] * s, dt);
    y[i] = rnd(a_lo + b_lo, dt);
    y[i + h] = rnd(a_hi + b_hi, dt);
  }
}

/*
 * Decode attention core for q_len == 1 (modify_llama.py:86-147).
 *   q [B,H,d]; kc/vc [B,Hkv,N,d] un-rotated cache INCLUDING the new row (the torch.cat of :95-98 is the
 *   caller's memcpy); cos/sin [>=max(N,pos_q+1), d/2]; mask [B,N] or NULL; out [B,H*d]; stash [B,H,N] or NULL.
 */
void orc_attn_decode(int dt, const void* q, const void* kc, const void* vc, const void* cos, const void* sin,
                     const void* mask, void* out, void* stash, int B, int H, int Hkv, int d, int N, int pos_q) {
  const float sqrt_d = sqrtf((float)d);
  const int grp = H / Hkv;
#pragma omp parallel
  {
    float* qx = (float*)malloc(sizeof(float) * d * 4);
    float *qr = qx + d, *kx = qx + 2 * d, *kr = qx + 3 * d;
    float* s = (float*)malloc(sizeof(float) * (size_t)N);
    float* acc = (float*)malloc(sizeof(float) * d);
#pragma omp for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < H; ++h) {
        const int hkv = h / grp;
        for (int i = 0; i < d; ++i) qx[i] = ld(q, ((int64_t)b * H + h) * d + i, dt);
        rope_row(qx, qr, cos, sin, pos_q, d, dt);                                   /* :92 */
        const int64_t kvb = ((int64_t)b * Hkv + hkv) * N * d;
        float m = -INFINITY;
        for (int j = 0; j < N; ++j) {
          for (int i = 0; i < d; ++i) kx[i] = ld(kc, kvb + (int64_t)j * d + i, dt);
          rope_row(kx, kr, cos, sin, j, d, dt);                                     /* :103-104, every step */
          float a = 0.f;
          for (int i = 0; i < d; ++i) a += qr[i] * kr[i];
          float sj = rnd(rnd(a, dt) / sqrt_d, dt);                                  /* :111-113 */
          if (stash) st(stash, ((int64_t)b * H + h) * N + j, sj, dt);               /* :116-119 */
          if (mask) sj = rnd(sj + ld(mask, (int64_t)b * N + j, dt), dt);            /* :132 */
          s[j] = sj;
          if (sj > m) m = sj;
        }
        float l = 0.f;
        for (int j = 0; j < N; ++j) { s[j] = expf(s[j] - m); l += s[j]; }            /* :135 fp32 softmax */
        for (int i = 0; i < d; ++i) acc[i] = 0.f;
        for (int j = 0; j < N; ++j) {
          const float p = rnd(s[j] / l, dt);                                        /* :135-137 .to(dtype) */
          const int64_t vb = kvb + (int64_t)j * d;
          for (int i = 0; i < d; ++i) acc[i] += p * ld(vc, vb + i, dt);             /* :138 */
        }
        for (int i = 0; i < d; ++i) st(out, ((int64_t)b * H + h) * d + i, acc[i], dt);   /* :146-147 layout */
      }
    free(qx); free(s); free(acc);
  }
}

/* order-preserving key: larger value => larger key, NaN largest, -0 == +0 (torch.topk order) */
static inline uint32_t okey(float x) {
  if (x != x) return 0xffffffffu;
  if (x == 0.0f) x = 0.0f;
  const uint32_t u = f2u(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
typedef struct { uint32_t key; int32_t idx; } orc_kv;
static int cmp_kv(const void* a, const void* b) {   /* key descending, index ascending */
  const orc_kv *x = (const orc_kv*)a, *y = (const orc_kv*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int cmp_i32(const void* a, const void* b) { return *(const int32_t*)a - *(const int32_t*)b; }

/* kv_cache_token_pruning.py:59-63: per head top-k of score[h, lo:hi), ascending positions (+lo).
 * Ties at the k-th value: lowest index first.  score [H, L] (row stride L).  Returns 0, or -1 if hi-lo < k. */
int orc_topk_window(int dt, const void* score, int H, int L, int lo, int hi, int k, int32_t* idx) {
  if (hi > L) hi = L;
  if (lo < 0 || k <= 0 || hi - lo < k) return -1;
  const int W = hi - lo;
#pragma omp parallel
  {
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)W);
#pragma omp for schedule(static)
    for (int h = 0; h < H; ++h) {
      for (int i = 0; i < W; ++i) { kv[i].key = okey(ld(score, (int64_t)h * L + lo + i, dt)); kv[i].idx = lo + i; }
      qsort(kv, (size_t)W, sizeof(orc_kv), cmp_kv);
      for (int i = 0; i < k; ++i) idx[(int64_t)h * k + i] = kv[i].idx;
      qsort(idx + (int64_t)h * k, (size_t)k, sizeof(int32_t), cmp_i32);
    }
    free(kv);
  }
  return 0;
}

/* kv_cache_token_pruning.py:64-96: gather of the kept rows + concat [start | important | tail].
 * src [B,H,L,d] -> dst [B,H,Lp,d], Lp = start + k + (L - tail_lo). */
void orc_kv_compact(int dt, const void* src, void* dst, const int32_t* idx, int B, int H, int L, int d, int start,
                    int k, int tail_lo) {
  if (tail_lo > L) tail_lo = L;
  const int tail = L - tail_lo, Lp = start + k + tail;
  const size_t rb = (size_t)d * esz(dt);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      const char* s = (const char*)src + ((size_t)b * H + h) * L * rb;
      char* o = (char*)dst + ((size_t)b * H + h) * Lp * rb;
      memcpy(o, s, rb * start);
      for (int i = 0; i < k; ++i) memcpy(o + rb * (start + i), s + rb * idx[(int64_t)h * k + i], rb);
      memcpy(o + rb * (start + k), s + rb * tail_lo, rb * tail);
    }
}
