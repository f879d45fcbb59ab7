// decode_attn.hip — fused single-token attention over an un-rotated, per-head-pruned KV cache.
//
// Replaces modify_llama.py:86-147 at q_len == 1 (reference = ~25 eager torch ops per layer):
//   RoPE(Q @ pos_q) · RoPE(K_j @ j)^T / sqrt(d)  -> stash (pre-mask)  -> +mask -> fp32 softmax -> P·V
// with the new token's K/V row appended in place (replaces torch.cat, :95-98).
//
// Shape of the problem (C2: B=1, H=32, d=128, n=2048 kept rows, bf16): 32 MiB of K+V per layer, 67 MFLOP
// -> HBM-bound (1 FLOP/B).  No MFMA: a 1xd by dxn product has no reuse to feed a matrix core with.
//
// Mapping (wave64, 256-thread workgroups):
//   grid = B*H*S workgroups; workgroup (b,h,s) owns `chunk` consecutive keys of head h (split-N so that
//   32 heads still fill 256 CUs).  A row of d elements is covered by LPR = d/16 lanes; lane c of a row
//   holds elements [8c,8c+8) and [d/2+8c, d/2+8c+8) — the two halves RoPE pairs up — so the rotation
//   needs no cross-lane traffic, and every global access is a 16-byte load of a fully used 128-byte
//   line (K/V rows are contiguous, pitch d).
//   A tile = UNR row-groups; all K, V, cos, sin loads of a tile are issued before the first use so one
//   workgroup keeps ~64 KiB of HBM reads in flight; 2-3 workgroups per CU overlap compute with loads.
//   Softmax is online across tiles (running max / sum), partial (o, m, l) per split goes to a small fp32
//   workspace; the LAST split to arrive for a (b,h) merges the partials (ticket counter, write-through
//   stores + agent-scope loads, no second launch).
//
// Rounding: in the 16-bit dtypes the reference rounds after every torch op; the kernel reproduces those
// roundings for everything that feeds the stash (x*cos, rot*sin, their sum, matmul result, /sqrt(d)) so
// stash values — the input of the top-k — match the reference except where fp32 accumulation ORDER
// moves a value across a rounding boundary.  P is NOT rounded to the model dtype before P·V (it would
// need the global softmax denominator before the first V row); documented tolerance in tests.
#include "common.h"

namespace spatten {

template <typename T>
struct DecodeParams {
  const T* q; int64_t q_sb, q_sh;
  T* kc; T* vc; int64_t kv_sb, kv_sh;
  const T* k_new; const T* v_new; int64_t new_sb, new_sh;
  const T* cos; const T* sin; int table_rows;
  const int64_t* pos_ids; int64_t pos_sb;
  const T* mask; int64_t mask_sb;
  T* out; int64_t out_sb;
  T* scores; int64_t sc_sb, sc_sh;
  float* lse;
  float* ws_part;       // [B*H, S, D+2]
  unsigned* ws_cnt;     // [B*H]
  int B, H, Hkv, N, pos_q, S, chunk;
  float sqrt_d;
};

constexpr int kDecodeThreads = 256;
// row-groups per tile: 4 for the 16-bit dtypes (6 x 16 B in flight per row-group and lane), 2 for fp32
template <typename T> constexpr int decode_unr() { return sizeof(T) == 4 ? 2 : 4; }

template <typename T, int D>
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

template <typename T, int D>
__global__ __launch_bounds__(kDecodeThreads) void decode_attn_kernel(const DecodeParams<T> p) {
  constexpr int LPR = D / 16;                    // lanes per row
  constexpr int RPI = kDecodeThreads / LPR;      // rows per iteration of the workgroup
  constexpr int UNR = decode_unr<T>();
  constexpr int TILE = RPI * UNR;
  constexpr int HALF = D / 2;
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;

  __shared__ float s_stash[TILE];
  __shared__ float s_red[4];
  __shared__ float s_o[4][D + 1];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x;
  const int c = tid % LPR;
  const int r = tid / LPR;
  const int wave = tid / kWave;
  const int lane = tid % kWave;

  const int split = blockIdx.x % p.S;
  const int bh = blockIdx.x / p.S;
  const int h = bh % p.H;
  const int b = bh / p.H;
  const int hkv = h / (p.H / p.Hkv);

  const int lo = split * p.chunk;
  const int hi = min(lo + p.chunk, p.N);

  // ---- rotated query (registers): elements [8c,8c+8) and [HALF+8c, HALF+8c+8) --------------------
  float qlo[8], qhi[8];
  {
    const T* qp = p.q + b * p.q_sb + h * p.q_sh;
    float xlo[8], xhi[8], cc[8], ss[8];
    V8::unpack(V8::ldg(qp + 8 * c), xlo);
    V8::unpack(V8::ldg(qp + HALF + 8 * c), xhi);
    int pq = p.pos_ids ? (int)p.pos_ids[b * p.pos_sb] : p.pos_q;
    pq = min(max(pq, 0), p.table_rows - 1);
    V8::unpack(V8::ldg(p.cos + (int64_t)pq * HALF + 8 * c), cc);
    V8::unpack(V8::ldg(p.sin + (int64_t)pq * HALF + 8 * c), ss);
    rope_pair<T, D>(xlo, xhi, cc, ss, qlo, qhi);
  }

  T* kbase = p.kc + b * p.kv_sb + hkv * p.kv_sh;
  T* vbase = p.vc + b * p.kv_sb + hkv * p.kv_sh;
  const bool has_new = (p.k_new != nullptr);

  float m_run = -INFINITY, l_run = 0.f;
  float olo[8], ohi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = 0.f; ohi[i] = 0.f; }

  for (int t0 = lo; t0 < hi; t0 += TILE) {
    // ---- issue every load of the tile -----------------------------------------------------------
    raw_t k_lo[UNR], k_hi[UNR], v_lo[UNR], v_hi[UNR], cs[UNR], sn[UNR];
    bool valid[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      int j = t0 + u * RPI + r;
      valid[u] = j < hi;
      j = valid[u] ? j : hi - 1;
      const T* kp = kbase + (int64_t)j * D;
      const T* vp = vbase + (int64_t)j * D;
      if (has_new && j == p.N - 1) {   // the token being appended: source = k_new / v_new
        kp = p.k_new + b * p.new_sb + hkv * p.new_sh;
        vp = p.v_new + b * p.new_sb + hkv * p.new_sh;
      }
      k_lo[u] = V8::ldg(kp + 8 * c);
      k_hi[u] = V8::ldg(kp + HALF + 8 * c);
      cs[u] = V8::ldg(p.cos + (int64_t)j * HALF + 8 * c);
      sn[u] = V8::ldg(p.sin + (int64_t)j * HALF + 8 * c);
      v_lo[u] = V8::ldg(vp + 8 * c);
      v_hi[u] = V8::ldg(vp + HALF + 8 * c);
    }
    if (has_new && t0 + TILE >= p.N && hi == p.N) {
      // append in place (modify_llama.py:95-100; K stays un-rotated)
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = t0 + u * RPI + r;
        if (j == p.N - 1) {
          V8::stg(kbase + (int64_t)j * D + 8 * c, k_lo[u]);
          V8::stg(kbase + (int64_t)j * D + HALF + 8 * c, k_hi[u]);
          V8::stg(vbase + (int64_t)j * D + 8 * c, v_lo[u]);
          V8::stg(vbase + (int64_t)j * D + HALF + 8 * c, v_hi[u]);
        }
      }
    }

    // ---- scores ---------------------------------------------------------------------------------
    float sc[UNR];
    float m_loc = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      V8::unpack(k_lo[u], xlo);
      V8::unpack(k_hi[u], xhi);
      V8::unpack(cs[u], cc);
      V8::unpack(sn[u], ss);
      rope_pair<T, D>(xlo, xhi, cc, ss, ylo, yhi);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(qlo[i], ylo[i], acc);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(qhi[i], yhi[i], acc);
      acc = group_sum<LPR>(acc);
      // matmul result -> dtype, then the separate divide -> dtype (modify_llama.py:111-113)
      float s = DT<T>::round(DT<T>::round(acc) / p.sqrt_d);
      const int j = t0 + u * RPI + r;
      if (c == 0) s_stash[u * RPI + r] = s;
      if (p.mask != nullptr && valid[u]) s = DT<T>::round(s + DT<T>::to_f32(p.mask[b * p.mask_sb + j]));
      sc[u] = valid[u] ? s : -INFINITY;
      m_loc = fmaxf(m_loc, sc[u]);
    }
    m_loc = wave_max(m_loc);
    if (lane == 0) s_red[wave] = m_loc;
    __syncthreads();
    // stash (raw scaled logits, pre-mask), coalesced
    if (p.scores != nullptr) {
      for (int i = tid; i < TILE; i += kDecodeThreads) {
        const int j = t0 + i;
        if (j < hi) p.scores[b * p.sc_sb + h * p.sc_sh + j] = DT<T>::from_f32(s_stash[i]);
      }
    }
    const float m_tile = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float m_new = fmaxf(m_run, m_tile);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __expf(m_run - m_use);     // m_run = -inf -> 0
    l_run *= alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) { olo[i] *= alpha; ohi[i] *= alpha; }
    m_run = m_new;

    // ---- P·V ------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float pj = __expf(sc[u] - m_use);      // invalid rows: exp(-inf) = 0
      l_run += pj;
      float vlo[8], vhi[8];
      V8::unpack(v_lo[u], vlo);
      V8::unpack(v_hi[u], vhi);
#pragma unroll
      for (int i = 0; i < 8; ++i) { olo[i] = fmaf(pj, vlo[i], olo[i]); ohi[i] = fmaf(pj, vhi[i], ohi[i]); }
    }
    __syncthreads();   // s_stash / s_red reused by the next tile
  }

  // ---- reduce over the row groups of the workgroup ----------------------------------------------
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    l_run += __shfl_xor(l_run, off, kWave);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      olo[i] += __shfl_xor(olo[i], off, kWave);
      ohi[i] += __shfl_xor(ohi[i], off, kWave);
    }
  }
  if (lane < LPR) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_o[wave][8 * lane + i] = olo[i];
      s_o[wave][HALF + 8 * lane + i] = ohi[i];
    }
    if (lane == 0) s_o[wave][D] = l_run;
  }
  __syncthreads();
  float o_tot = 0.f, l_tot = 0.f;
  if (tid < D) o_tot = s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid];
  l_tot = s_o[0][D] + s_o[1][D] + s_o[2][D] + s_o[3][D];

  if (p.S == 1) {
    if (tid < D) p.out[b * p.out_sb + h * D + tid] = DT<T>::from_f32(o_tot / l_tot);
    if (p.lse != nullptr && tid == 0) { p.lse[bh * 2] = m_run; p.lse[bh * 2 + 1] = l_tot; }
    return;
  }

  // ---- publish the partial; the last split to arrive merges ------------------------------------
  // Write-through (agent-scope relaxed atomic = sc1) stores + drained counter; the merger reads with
  // agent-scope loads: placement independent across the 8 XCD L2s, no fences needed.
  float* part = p.ws_part + ((int64_t)bh * p.S + split) * (D + 2);
  if (tid < D) __hip_atomic_store(part + tid, o_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    __hip_atomic_store(part + D, m_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + D + 1, l_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.ws_cnt + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != (unsigned)(p.S - 1)) return;

  const float* base = p.ws_part + (int64_t)bh * p.S * (D + 2);
  float m_g = -INFINITY;
  for (int s = 0; s < p.S; ++s)
    m_g = fmaxf(m_g, __hip_atomic_load(base + s * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const float m_gu = (m_g == -INFINITY) ? 0.f : m_g;
  float l_g = 0.f, o_g = 0.f;
  for (int s = 0; s < p.S; ++s) {
    const float ms = __hip_atomic_load(base + s * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float ls = __hip_atomic_load(base + s * (D + 2) + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float w = __expf(ms - m_gu);
    l_g = fmaf(ls, w, l_g);
    if (tid < D) o_g = fmaf(__hip_atomic_load(base + s * (D + 2) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w, o_g);
  }
  if (tid < D) p.out[b * p.out_sb + h * D + tid] = DT<T>::from_f32(o_g / l_g);
  if (tid == 0) {
    if (p.lse != nullptr) { p.lse[bh * 2] = m_g; p.lse[bh * 2 + 1] = l_g; }
    __hip_atomic_store(p.ws_cnt + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int decode_tile_rows(int d, int dtype) {
  return (kDecodeThreads / (d / 16)) * (dtype == SPATTEN_F32 ? 2 : 4);
}

static int auto_splits(int batch, int heads, int d, int kv_len, int dtype = SPATTEN_BF16) {
  const int tile = decode_tile_rows(d, dtype);
  const int max_by_len = ceil_div(kv_len, tile);
  // aim for >= 4 workgroups per CU-slot pair: 256 CUs x 2
  int s = ceil_div(512, batch * heads);
  if (s > max_by_len) s = max_by_len;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}

template <typename T, int D>
static int launch_decode(const DecodeParams<T>& p, hipStream_t stream) {
  const dim3 grid((unsigned)(p.B * p.H * p.S));
  hipLaunchKernelGGL((decode_attn_kernel<T, D>), grid, dim3(kDecodeThreads), 0, stream, p);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

template <typename T>
static int dispatch_decode(DecodeParams<T>& p, int d, hipStream_t stream) {
  switch (d) {
    case 64: return launch_decode<T, 64>(p, stream);
    case 128: return launch_decode<T, 128>(p, stream);
    case 256: return launch_decode<T, 256>(p, stream);
    default: return SPATTEN_ERR_UNSUPPORTED;
  }
}

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_decode_workspace_bytes(int batch, int heads, int head_dim, int max_splits) {
  if (batch <= 0 || heads <= 0 || head_dim <= 0 || max_splits <= 0) return 0;
  const size_t cnt = ((size_t)batch * heads * sizeof(unsigned) + 255) / 256 * 256;
  return cnt + (size_t)batch * heads * max_splits * (head_dim + 2) * sizeof(float);
}

extern "C" int spatten_decode_auto_splits(int batch, int heads, int head_dim, int kv_len) {
  if (batch <= 0 || heads <= 0 || kv_len <= 0 || (head_dim != 64 && head_dim != 128 && head_dim != 256)) return 1;
  return auto_splits(batch, heads, head_dim, kv_len);
}

extern "C" int spatten_attn_decode(int dtype, const void* q, int64_t q_sb, int64_t q_sh, void* k_cache,
                                   void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* k_new,
                                   const void* v_new, int64_t new_sb, int64_t new_sh, const void* cos,
                                   const void* sin, int table_rows, const int64_t* position_ids,
                                   int64_t pos_sb, const void* mask, int64_t mask_sb, void* out,
                                   int64_t out_sb, void* scores, int64_t sc_sb, int64_t sc_sh, float* lse,
                                   void* workspace, int batch, int heads, int kv_heads, int head_dim,
                                   int kv_len, int pos_q, int n_splits, void* stream) {
  if (!q || !k_cache || !v_cache || !cos || !sin || !out) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || kv_len <= 0 || pos_q < 0)
    return SPATTEN_ERR_INVALID;
  if (table_rows < kv_len || (!position_ids && pos_q >= table_rows)) return SPATTEN_ERR_INVALID;
  if ((k_new == nullptr) != (v_new == nullptr)) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128 && head_dim != 256) return SPATTEN_ERR_UNSUPPORTED;
  const int tile = decode_tile_rows(head_dim, dtype);
  int S = n_splits > 0 ? n_splits : auto_splits(batch, heads, head_dim, kv_len, dtype);
  if (S > ceil_div(kv_len, tile)) S = ceil_div(kv_len, tile);
  if (S > 1 && !workspace) return SPATTEN_ERR_INVALID;
  // chunk = rows per split, a multiple of the tile so every split starts tile-aligned
  int chunk = ceil_div(ceil_div(kv_len, S), tile) * tile;
  S = ceil_div(kv_len, chunk);
  const size_t cnt_bytes = ((size_t)batch * heads * sizeof(unsigned) + 255) / 256 * 256;

#define SPATTEN_FILL(T)                                                                                  \
  DecodeParams<T> p;                                                                                     \
  p.q = (const T*)q; p.q_sb = q_sb; p.q_sh = q_sh;                                                       \
  p.kc = (T*)k_cache; p.vc = (T*)v_cache; p.kv_sb = kv_sb; p.kv_sh = kv_sh;                              \
  p.k_new = (const T*)k_new; p.v_new = (const T*)v_new; p.new_sb = new_sb; p.new_sh = new_sh;            \
  p.cos = (const T*)cos; p.sin = (const T*)sin; p.table_rows = table_rows;                               \
  p.pos_ids = position_ids; p.pos_sb = pos_sb;                                                           \
  p.mask = (const T*)mask; p.mask_sb = mask_sb;                                                          \
  p.out = (T*)out; p.out_sb = out_sb;                                                                    \
  p.scores = (T*)scores; p.sc_sb = sc_sb; p.sc_sh = sc_sh;                                               \
  p.lse = lse;                                                                                           \
  p.ws_cnt = (unsigned*)workspace;                                                                       \
  p.ws_part = workspace ? (float*)((char*)workspace + cnt_bytes) : nullptr;                              \
  p.B = batch; p.H = heads; p.Hkv = kv_heads; p.N = kv_len; p.pos_q = pos_q; p.S = S; p.chunk = chunk;   \
  p.sqrt_d = sqrtf((float)head_dim);                                                                     \
  return dispatch_decode<T>(p, head_dim, (hipStream_t)stream);

  switch (dtype) {
    case SPATTEN_F32: { SPATTEN_FILL(float) }
    case SPATTEN_F16: { SPATTEN_FILL(f16_t) }
    case SPATTEN_BF16: { SPATTEN_FILL(bf16_t) }
    default: return SPATTEN_ERR_INVALID;
  }
#undef SPATTEN_FILL
}
