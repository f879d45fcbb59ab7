// prune.hip — the prune event of SpAttenKVCache.apply_token_pruning (kv_cache_token_pruning.py:42-96):
//   importance (:51) -> per-head window top-k, ascending positions (:59-63) -> mask gather + concat (:64-96)
//
// The reference does, per layer: torch.topk + sort + scatter into a bool mask + .cpu() + boolean gather +
// three-way cat (two full copies of the kept rows and a device->host sync per layer).  Here:
//   * topk_select_kernel: one workgroup per (layer, head).  Radix-select (8-bit digits, MSB first) on the
//     order-preserving integer key of the score finds the k-th largest; an order-preserving compaction
//     (wave ballot + popcount prefix, the ZeroEliminator/PrefixSum shape of the RTL) then emits the kept
//     positions already ASCENDING — no sort, no mask.  Ties at the threshold: lowest index first
//     (TopK.scala:193-212), exactly k outputs.
//   * kv_compact_kernel: one pass, HBM-bound.  Every 16-byte piece of every destination row of K and V (all
//     layers in one launch) is produced by one lane: start rows, gathered rows, tail rows.  Kept rows are
//     contiguous 2*d-byte bursts (row-major [.., L, d] layout), so both sides are fully coalesced.
#include "common.h"

namespace spatten {

// ================================================================================================
// importance = stash.sum(0).sum(1)      (kv_cache_token_pruning.py:51)
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void importance_kernel(const T* __restrict__ stash, int64_t sb, int64_t sh,
                                                         int64_t sq, T* __restrict__ out, int64_t out_sh,
                                                         int B, int H, int Q, int L) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y;
  if (j >= L) return;
  float acc_q = 0.f;
  for (int q = 0; q < Q; ++q) {
    float acc_b = 0.f;
    for (int b = 0; b < B; ++b) acc_b += DT<T>::to_f32(stash[b * sb + h * sh + q * sq + j]);
    acc_q += DT<T>::round(acc_b);          // .sum(0) rounds to the stash dtype
  }
  out[h * out_sh + j] = DT<T>::from_f32(acc_q);   // .sum(1) rounds again
}

// ================================================================================================
// window top-k -> ascending absolute positions
// ================================================================================================
constexpr int kSelThreads = 256;

template <typename T>
struct SelectParams {
  const T* score;                 // single-layer base, or
  const void* const* score_ptrs;  // per-layer pointers (device memory)
  int64_t score_sh;
  int32_t* idx;                   // [layers, H, k] via idx_sl, idx_sh
  int64_t idx_sl, idx_sh;
  int H, lo, hi, k;
};

template <typename T>
__global__ __launch_bounds__(kSelThreads) void topk_select_kernel(const SelectParams<T> p) {
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_sel[2];          // digit, count strictly above it
  __shared__ unsigned s_cnt[2][4][2];    // [parity][wave][gt, eq]

  const int tid = threadIdx.x;
  const int lane = tid % kWave;
  const int wave = tid / kWave;
  const int h = blockIdx.x % p.H;
  const int layer = blockIdx.x / p.H;
  const T* row = (p.score_ptrs ? (const T*)p.score_ptrs[layer] : p.score) + h * p.score_sh + p.lo;
  int32_t* out = p.idx + layer * p.idx_sl + h * p.idx_sh;
  const int W = p.hi - p.lo;

  // ---- radix select: the key of the k-th largest element ----------------------------------------
  // fp32 keys of bf16 values populate the top 16 bits, of f16 values the top 19 (1+8+10), fp32 all 32.
  // The radix passes only look at those bits, so the key is masked to them everywhere (the all-ones NaN key
  // would otherwise compare GREATER than the threshold assembled from the passes and over-fill the output).
  constexpr int kPasses = sizeof(T) == 4 ? 4 : (DT<T>::kId == SPATTEN_BF16 ? 2 : 3);
  constexpr unsigned kKeyMask = kPasses == 4 ? 0xFFFFFFFFu : (kPasses == 3 ? 0xFFFFFF00u : 0xFFFF0000u);
  unsigned prefix = 0, pmask = 0;
  unsigned k_rem = (unsigned)p.k;
#pragma unroll 1
  for (int pass = 0; pass < kPasses; ++pass) {
    const int shift = 24 - 8 * pass;
    s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < W; i += kSelThreads) {
      const unsigned key = ordered_key(DT<T>::to_f32(row[i])) & kKeyMask;
      if ((key & pmask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      // lane l owns the 4 bins 255-4l .. 252-4l (descending); inclusive scan over lanes
      unsigned c[4], tot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c[j] = s_hist[255 - 4 * lane - j]; tot += c[j]; }
      unsigned inc = tot;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned t = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += t;
      }
      unsigned ex = inc - tot;
      if (ex < k_rem && k_rem <= inc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ex < k_rem && k_rem <= ex + c[j]) { s_sel[0] = 255 - 4 * lane - j; s_sel[1] = ex; }
          ex += c[j];
        }
      }
    }
    __syncthreads();
    prefix |= s_sel[0] << shift;
    pmask |= 255u << shift;
    k_rem -= s_sel[1];
  }
  const unsigned thr = prefix;       // key of the k-th largest
  const unsigned need_eq = k_rem;    // how many elements equal to it are kept (>= 1), lowest index first

  // ---- order-preserving compaction --------------------------------------------------------------
  unsigned run_eq = 0, run_kept = 0;
  int parity = 0;
  for (int base = 0; base < W; base += kSelThreads, parity ^= 1) {
    const int i = base + tid;
    unsigned key = 0;
    const bool in = i < W;
    if (in) key = ordered_key(DT<T>::to_f32(row[i])) & kKeyMask;
    const bool gt = in && key > thr;
    const bool eq = in && key == thr;
    const unsigned long long m_gt = __ballot(gt);
    const unsigned long long m_eq = __ballot(eq);
    if (lane == 0) { s_cnt[parity][wave][0] = __popcll(m_gt); s_cnt[parity][wave][1] = __popcll(m_eq); }
    __syncthreads();
    unsigned eq_base = run_eq, kept_base = run_kept;
    unsigned tot_eq = run_eq, tot_kept = run_kept;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned g = s_cnt[parity][w][0], e = s_cnt[parity][w][1];
      const unsigned room = tot_eq < need_eq ? need_eq - tot_eq : 0u;
      const unsigned kept_w = g + (e < room ? e : room);
      if (w < wave) { eq_base += e; kept_base += kept_w; }
      tot_eq += e;
      tot_kept += kept_w;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned my_eq_rank = eq_base + __popcll(m_eq & lt);
    const bool keep = gt || (eq && my_eq_rank < need_eq);
    const unsigned long long m_keep = __ballot(keep);
    const unsigned pos = kept_base + __popcll(m_keep & lt);
    if (keep && pos < (unsigned)p.k) out[pos] = p.lo + i;      // pos < k always holds; the guard keeps a logic slip local
    run_eq = tot_eq;
    run_kept = tot_kept;
  }
}

// ================================================================================================
// fused gather + concat of K and V, all layers
// ================================================================================================
constexpr int kCompThreads = 256;

struct CompactParams {
  const void* k_src; const void* v_src; void* k_dst; void* v_dst; void* kr_dst;   // single layer, or
  const void* const* k_src_ptrs; const void* const* v_src_ptrs;                    // per-layer pointers (device memory)
  void* const* k_dst_ptrs; void* const* v_dst_ptrs; void* const* kr_dst_ptrs;
  const void* cos; const void* sin; int table_rows;   // rotary half tables, only for the shadow (kr) output
  int64_t src_sb, src_sh, dst_sb, dst_sh;   // in BYTES
  const int32_t* idx; int64_t idx_sl, idx_sh;
  int B, H, layers, n_tensors;              // n_tensors: 2 = K and V, 1 = K only
  int start, k, tail_lo, Lp;                // Lp = start + k + tail_len
  int ppr;                                  // 16-byte pieces per row
  int row_bytes;
  int hp_shift;                             // log2(ppr / 2) when that is a power of two, else -1
  int rows_per_block;                       // kCompThreads / (ppr / 2)
};

// One work item = the PAIR of 16-byte pieces (pc, pc + ppr/2) of one destination row — the two halves RoPE
// pairs up — so the optional rotated-shadow output (row r of the new cache rotated at its NEW slot index r,
// modify_llama.py:103-104) is produced in registers by the lane that moves the row: the shadow is rebuilt by
// the same pass that moves the rows, and every lane of a wave does the same amount of work.
// Grid = (row blocks, B*H, layers * planes): everything a lane needs beyond its row and piece is wave-uniform (scalar
// ALU), and row / piece come out of the thread index by shift and mask — r02: the first version decoded a flat work
// index with four integer divisions per 32 bytes moved and was bound by that arithmetic, not by HBM; one item per lane
// (no unrolling: the registers buy occupancy, which a pure gather needs more than ILP) measured 500 vs 542 us.
template <typename T>
__global__ __launch_bounds__(kCompThreads) void kv_compact_kernel(const CompactParams p) {
  const int half_ppr = p.ppr >> 1;
  const int half_bytes = p.row_bytes >> 1;
  int piece, rloc;
  if (p.hp_shift >= 0) { piece = threadIdx.x & (half_ppr - 1); rloc = threadIdx.x >> p.hp_shift; }
  else { rloc = threadIdx.x / half_ppr; piece = threadIdx.x - rloc * half_ppr; }
  const int r = blockIdx.x * p.rows_per_block + rloc;
  if (rloc >= p.rows_per_block || r >= p.Lp) return;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;                      // wave-uniform
  const int layer = blockIdx.z / p.n_tensors, t = blockIdx.z - layer * p.n_tensors;
  const bool want_kr = t == 0 && ((p.kr_dst != nullptr) || (p.kr_dst_ptrs != nullptr));
  int src_row;
  if (r < p.start) src_row = r;
  else if (r < p.start + p.k) src_row = p.idx[layer * p.idx_sl + h * p.idx_sh + (r - p.start)];
  else src_row = p.tail_lo + (r - p.start - p.k);
  const char* sbase;
  char* dbase;
  if (p.k_src_ptrs) {
    sbase = (const char*)(t == 0 ? p.k_src_ptrs[layer] : p.v_src_ptrs[layer]);
    dbase = (char*)(t == 0 ? p.k_dst_ptrs[layer] : p.v_dst_ptrs[layer]);
  } else {
    sbase = (const char*)(t == 0 ? p.k_src : p.v_src);
    dbase = (char*)(t == 0 ? p.k_dst : p.v_dst);
  }
  const char* sp = sbase + b * p.src_sb + h * p.src_sh + (int64_t)src_row * p.row_bytes + piece * 16;
  const int64_t doff = b * p.dst_sb + h * p.dst_sh + (int64_t)r * p.row_bytes + piece * 16;
  const u32x4 lo_v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp));
  const u32x4 hi_v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp + half_bytes));
  u32x4 cs, sn;
  if (want_kr) {
    const int pos = min(r, p.table_rows - 1);
    const int64_t toff = (int64_t)pos * half_bytes + piece * 16;
    cs = *reinterpret_cast<const u32x4*>((const char*)p.cos + toff);
    sn = *reinterpret_cast<const u32x4*>((const char*)p.sin + toff);
  }
  __builtin_nontemporal_store(lo_v, reinterpret_cast<u32x4*>(dbase + doff));
  __builtin_nontemporal_store(hi_v, reinterpret_cast<u32x4*>(dbase + doff + half_bytes));
  if (want_kr) {
    char* rbase = (char*)(p.kr_dst_ptrs ? p.kr_dst_ptrs[layer] : p.kr_dst);
    constexpr int E = 16 / sizeof(T);              // elements per 16-byte piece (8, or 4 for fp32)
    const T* xl = reinterpret_cast<const T*>(&lo_v);
    const T* xh = reinterpret_cast<const T*>(&hi_v);
    const T* cc = reinterpret_cast<const T*>(&cs);
    const T* ss = reinterpret_cast<const T*>(&sn);
    u32x4 olo, ohi;
    T* yl = reinterpret_cast<T*>(&olo);
    T* yh = reinterpret_cast<T*>(&ohi);
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float a = DT<T>::to_f32(xl[e]), bb = DT<T>::to_f32(xh[e]);
        const float c = DT<T>::to_f32(cc[e]), s_ = DT<T>::to_f32(ss[e]);
        yl[e] = DT<T>::from_f32(DT<T>::round(a * c) + DT<T>::round(-bb * s_));
        yh[e] = DT<T>::from_f32(DT<T>::round(bb * c) + DT<T>::round(a * s_));
      }
    }
    __builtin_nontemporal_store(olo, reinterpret_cast<u32x4*>(rbase + doff));
    __builtin_nontemporal_store(ohi, reinterpret_cast<u32x4*>(rbase + doff + half_bytes));
  }
}

// ================================================================================================
// apply_rotary_pos_emb_single (modify_llama.py:21-28)
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void rope_single_kernel(const T* __restrict__ x, int64_t x_sb, int64_t x_sh,
                                                          int64_t x_sn, T* __restrict__ y, int64_t y_sb,
                                                          int64_t y_sh, int64_t y_sn, const T* __restrict__ cos,
                                                          const T* __restrict__ sin, int table_rows,
                                                          const int64_t* __restrict__ pos, int64_t pos_sb,
                                                          int pos0, int B, int H, int n, int d) {
  // one thread per (b, h, row, group of 8 column pairs)
  const int half = d / 2;
  const int gpr = half / 8;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * H * n * gpr;
  if (g >= total) return;
  const int c = (int)(g % gpr);
  const long long rowid = g / gpr;
  const int i = (int)(rowid % n);
  const int h = (int)((rowid / n) % H);
  const int b = (int)(rowid / ((long long)n * H));
  int ps = pos ? (int)pos[b * pos_sb + i] : pos0 + i;
  ps = min(max(ps, 0), table_rows - 1);
  using V8 = Vec8<T>;
  const T* xp = x + b * x_sb + h * x_sh + (int64_t)i * x_sn;
  T* yp = y + b * y_sb + h * y_sh + (int64_t)i * y_sn;
  float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
  V8::unpack(V8::ldg(xp + 8 * c), xlo);
  V8::unpack(V8::ldg(xp + half + 8 * c), xhi);
  V8::unpack(V8::ldg(cos + (int64_t)ps * half + 8 * c), cc);
  V8::unpack(V8::ldg(sin + (int64_t)ps * half + 8 * c), ss);
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ylo[e] = DT<T>::round(DT<T>::round(xlo[e] * cc[e]) + DT<T>::round(-xhi[e] * ss[e]));
      yhi[e] = DT<T>::round(DT<T>::round(xhi[e] * cc[e]) + DT<T>::round(xlo[e] * ss[e]));
    }
  }
  V8::stg(yp + 8 * c, V8::pack(ylo));
  V8::stg(yp + half + 8 * c, V8::pack(yhi));
}

// ================================================================================================
// KV append without attention: rows [row0, row0+n) of k / v, and their rotation at the slot index into the shadow
// (modify_llama.py:95-104).  One thread per (b, h, row, group of 8 column pairs).
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* __restrict__ k_new, const T* __restrict__ v_new,
                                                        int64_t new_sb, int64_t new_sh, int64_t new_sn,
                                                        T* __restrict__ kc, T* __restrict__ krc, T* __restrict__ vc,
                                                        int64_t kv_sb, int64_t kv_sh, const T* __restrict__ cos,
                                                        const T* __restrict__ sin, int table_rows, int B, int H, int n,
                                                        int d, int row0) {
  const int half = d / 2;
  const int gpr = half / 8;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * H * n * gpr;
  if (g >= total) return;
  const int c = (int)(g % gpr);
  const long long rowid = g / gpr;
  const int i = (int)(rowid % n);
  const int h = (int)((rowid / n) % H);
  const int b = (int)(rowid / ((long long)n * H));
  const int ps = min(row0 + i, table_rows - 1);
  using V8 = Vec8<T>;
  const T* kp = k_new + b * new_sb + h * new_sh + (int64_t)i * new_sn;
  const T* vp = v_new + b * new_sb + h * new_sh + (int64_t)i * new_sn;
  const int64_t dst = b * kv_sb + h * kv_sh + (int64_t)(row0 + i) * d;
  const typename V8::raw k0 = V8::ldg(kp + 8 * c), k1 = V8::ldg(kp + half + 8 * c);
  const typename V8::raw v0 = V8::ldg(vp + 8 * c), v1 = V8::ldg(vp + half + 8 * c);
  float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
  V8::unpack(k0, xlo);
  V8::unpack(k1, xhi);
  V8::unpack(V8::ldg(cos + (int64_t)ps * half + 8 * c), cc);
  V8::unpack(V8::ldg(sin + (int64_t)ps * half + 8 * c), ss);
  rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
  if (kc) { V8::stg(kc + dst + 8 * c, k0); V8::stg(kc + dst + half + 8 * c, k1); }
  V8::stg(krc + dst + 8 * c, V8::pack(ylo));
  V8::stg(krc + dst + half + 8 * c, V8::pack(yhi));
  V8::stg(vc + dst + 8 * c, v0);
  V8::stg(vc + dst + half + 8 * c, v1);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int elt_size(int dtype) { return dtype == SPATTEN_F32 ? 4 : 2; }
static inline bool valid_dtype(int dtype) { return dtype == SPATTEN_F32 || dtype == SPATTEN_F16 || dtype == SPATTEN_BF16; }

template <typename T>
static int launch_select(const void* score, const void* const* score_ptrs, int64_t score_sh, int layers, int H,
                         int lo, int hi, int k, int32_t* idx, int64_t idx_sl, int64_t idx_sh, hipStream_t st) {
  SelectParams<T> p;
  p.score = (const T*)score; p.score_ptrs = score_ptrs; p.score_sh = score_sh;
  p.idx = idx; p.idx_sl = idx_sl; p.idx_sh = idx_sh;
  p.H = H; p.lo = lo; p.hi = hi; p.k = k;
  hipLaunchKernelGGL((topk_select_kernel<T>), dim3((unsigned)(layers * H)), dim3(kSelThreads), 0, st, p);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

static int select_any(int dtype, const void* score, const void* const* score_ptrs, int64_t score_sh, int layers,
                      int H, int lo, int hi, int k, int32_t* idx, int64_t idx_sl, int64_t idx_sh, hipStream_t st) {
  if (!valid_dtype(dtype) || (!score && !score_ptrs) || !idx || H <= 0 || layers <= 0) return SPATTEN_ERR_INVALID;
  if (lo < 0 || k <= 0 || hi - lo < k) return SPATTEN_ERR_WINDOW;
  switch (dtype) {
    case SPATTEN_F32: return launch_select<float>(score, score_ptrs, score_sh, layers, H, lo, hi, k, idx, idx_sl, idx_sh, st);
    case SPATTEN_F16: return launch_select<f16_t>(score, score_ptrs, score_sh, layers, H, lo, hi, k, idx, idx_sl, idx_sh, st);
    default: return launch_select<bf16_t>(score, score_ptrs, score_sh, layers, H, lo, hi, k, idx, idx_sl, idx_sh, st);
  }
}

static int compact_any(int dtype, CompactParams& p, int head_dim, int tail_len, hipStream_t st) {
  if (!valid_dtype(dtype) || p.B <= 0 || p.H <= 0 || head_dim <= 0 || p.layers <= 0) return SPATTEN_ERR_INVALID;
  if (p.start < 0 || p.k < 0 || tail_len < 0 || p.tail_lo < 0) return SPATTEN_ERR_INVALID;
  if (p.k > 0 && !p.idx) return SPATTEN_ERR_INVALID;
  const int es = elt_size(dtype);
  p.row_bytes = head_dim * es;
  if (p.row_bytes % 16 != 0) return SPATTEN_ERR_UNSUPPORTED;
  p.ppr = p.row_bytes / 16;
  p.Lp = p.start + p.k + tail_len;
  p.src_sb *= es; p.src_sh *= es; p.dst_sb *= es; p.dst_sh *= es;
  if (p.ppr % 2 != 0) return SPATTEN_ERR_UNSUPPORTED;      // rows are moved as (first half, second half) piece pairs
  const int half_ppr = p.ppr / 2;
  if (half_ppr > kCompThreads) return SPATTEN_ERR_UNSUPPORTED;
  p.hp_shift = -1;
  for (int sft = 0; sft < 9; ++sft) if ((1 << sft) == half_ppr) p.hp_shift = sft;
  p.rows_per_block = kCompThreads / half_ppr;
  if (p.Lp == 0) return SPATTEN_OK;
  const long long blocks_x = ((long long)p.Lp + p.rows_per_block - 1) / p.rows_per_block;
  const long long grid_y = (long long)p.B * p.H, grid_z = (long long)p.layers * p.n_tensors;
  if (blocks_x >= (1ll << 31) || grid_y > 65535 || grid_z > 65535) return SPATTEN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks_x, (unsigned)grid_y, (unsigned)grid_z);
  const bool want_kr = p.kr_dst || p.kr_dst_ptrs;
  if (want_kr && (!p.cos || !p.sin || p.table_rows < p.Lp)) return SPATTEN_ERR_INVALID;
  switch (dtype) {
    case SPATTEN_F32: hipLaunchKernelGGL((kv_compact_kernel<float>), grid, dim3(kCompThreads), 0, st, p); break;
    case SPATTEN_F16: hipLaunchKernelGGL((kv_compact_kernel<f16_t>), grid, dim3(kCompThreads), 0, st, p); break;
    default: hipLaunchKernelGGL((kv_compact_kernel<bf16_t>), grid, dim3(kCompThreads), 0, st, p);
  }
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

}  // namespace spatten

using namespace spatten;

extern "C" int spatten_importance(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq, void* out,
                                  int64_t out_sh, int batch, int heads, int q_len, int kv_len, void* stream) {
  if (!valid_dtype(dtype) || !stash || !out || batch <= 0 || heads <= 0 || q_len <= 0 || kv_len <= 0)
    return SPATTEN_ERR_INVALID;
  const dim3 grid((unsigned)ceil_div(kv_len, 256), (unsigned)heads);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case SPATTEN_F32:
      hipLaunchKernelGGL((importance_kernel<float>), grid, dim3(256), 0, st, (const float*)stash, sb, sh, sq,
                         (float*)out, out_sh, batch, heads, q_len, kv_len);
      break;
    case SPATTEN_F16:
      hipLaunchKernelGGL((importance_kernel<f16_t>), grid, dim3(256), 0, st, (const f16_t*)stash, sb, sh, sq,
                         (f16_t*)out, out_sh, batch, heads, q_len, kv_len);
      break;
    default:
      hipLaunchKernelGGL((importance_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)stash, sb, sh, sq,
                         (bf16_t*)out, out_sh, batch, heads, q_len, kv_len);
  }
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_topk_select(int dtype, const void* score, int64_t score_sh, int heads, int lo, int hi,
                                   int k, int32_t* idx, int64_t idx_sh, void* stream) {
  return select_any(dtype, score, nullptr, score_sh, 1, heads, lo, hi, k, idx, 0, idx_sh, (hipStream_t)stream);
}

extern "C" int spatten_kv_compact(int dtype, const void* k_src, const void* v_src, int64_t src_sb, int64_t src_sh,
                                  void* k_dst, void* v_dst, void* kr_dst, int64_t dst_sb, int64_t dst_sh,
                                  const void* cos, const void* sin, int table_rows, const int32_t* idx,
                                  int64_t idx_sh, int batch, int heads, int head_dim, int start, int k,
                                  int tail_lo, int tail_len, void* stream) {
  if (!k_src || !k_dst || ((v_src == nullptr) != (v_dst == nullptr))) return SPATTEN_ERR_INVALID;
  CompactParams p{};
  p.k_src = k_src; p.v_src = v_src; p.k_dst = k_dst; p.v_dst = v_dst; p.kr_dst = kr_dst;
  p.cos = cos; p.sin = sin; p.table_rows = table_rows;
  p.src_sb = src_sb; p.src_sh = src_sh; p.dst_sb = dst_sb; p.dst_sh = dst_sh;
  p.idx = idx; p.idx_sl = 0; p.idx_sh = idx_sh;
  p.B = batch; p.H = heads; p.layers = 1; p.n_tensors = v_src ? 2 : 1;
  p.start = start; p.k = k; p.tail_lo = tail_lo;
  return compact_any(dtype, p, head_dim, tail_len, (hipStream_t)stream);
}

extern "C" int spatten_prune_layers(int dtype, int layers, const void* const* score_ptrs, int64_t score_sh,
                                    const void* const* k_src_ptrs, const void* const* v_src_ptrs, int64_t src_sb,
                                    int64_t src_sh, void* const* k_dst_ptrs, void* const* v_dst_ptrs,
                                    void* const* kr_dst_ptrs, int64_t dst_sb, int64_t dst_sh, const void* cos,
                                    const void* sin, int table_rows, int32_t* idx, int batch, int heads,
                                    int head_dim, int lo, int hi, int k, int tail_lo, int tail_len, void* stream) {
  if (!score_ptrs || !k_src_ptrs || !v_src_ptrs || !k_dst_ptrs || !v_dst_ptrs || !idx) return SPATTEN_ERR_INVALID;
  const int rc = select_any(dtype, nullptr, score_ptrs, score_sh, layers, heads, lo, hi, k, idx,
                            (int64_t)heads * k, k, (hipStream_t)stream);
  if (rc != SPATTEN_OK) return rc;
  CompactParams p{};
  p.k_src_ptrs = k_src_ptrs; p.v_src_ptrs = v_src_ptrs; p.k_dst_ptrs = k_dst_ptrs; p.v_dst_ptrs = v_dst_ptrs;
  p.kr_dst_ptrs = kr_dst_ptrs; p.cos = cos; p.sin = sin; p.table_rows = table_rows;
  p.src_sb = src_sb; p.src_sh = src_sh; p.dst_sb = dst_sb; p.dst_sh = dst_sh;
  p.idx = idx; p.idx_sl = (int64_t)heads * k; p.idx_sh = k;
  p.B = batch; p.H = heads; p.layers = layers; p.n_tensors = 2;
  p.start = lo; p.k = k; p.tail_lo = tail_lo;
  return compact_any(dtype, p, head_dim, tail_len, (hipStream_t)stream);
}

extern "C" int spatten_rope_single(int dtype, const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sn, void* y,
                                   int64_t y_sb, int64_t y_sh, int64_t y_sn, const void* cos, const void* sin,
                                   int table_rows, const int64_t* position_ids, int64_t pos_sb, int pos0,
                                   int batch, int heads, int n, int head_dim, void* stream) {
  if (!valid_dtype(dtype) || !x || !y || !cos || !sin || batch <= 0 || heads <= 0 || n <= 0 || table_rows <= 0)
    return SPATTEN_ERR_INVALID;
  if (head_dim <= 0 || head_dim % 16 != 0) return SPATTEN_ERR_UNSUPPORTED;
  const long long total = (long long)batch * heads * n * (head_dim / 16);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define SPATTEN_ROPE(T)                                                                                         \
  hipLaunchKernelGGL((rope_single_kernel<T>), grid, dim3(256), 0, st, (const T*)x, x_sb, x_sh, x_sn, (T*)y, y_sb, \
                     y_sh, y_sn, (const T*)cos, (const T*)sin, table_rows, position_ids, pos_sb, pos0, batch,    \
                     heads, n, head_dim)
  switch (dtype) {
    case SPATTEN_F32: SPATTEN_ROPE(float); break;
    case SPATTEN_F16: SPATTEN_ROPE(f16_t); break;
    default: SPATTEN_ROPE(bf16_t);
  }
#undef SPATTEN_ROPE
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_kv_append(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh,
                                 int64_t new_sn, void* k_cache, void* kr_cache, void* v_cache, int64_t kv_sb,
                                 int64_t kv_sh, const void* cos, const void* sin, int table_rows, int batch,
                                 int kv_heads, int n, int head_dim, int row0, void* stream) {
  if (!valid_dtype(dtype) || !k_new || !v_new || !kr_cache || !v_cache || !cos || !sin) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || kv_heads <= 0 || n < 0 || row0 < 0 || table_rows < row0 + n) return SPATTEN_ERR_INVALID;
  if (head_dim <= 0 || head_dim % 16 != 0) return SPATTEN_ERR_UNSUPPORTED;
  if (n == 0) return SPATTEN_OK;
  const long long total = (long long)batch * kv_heads * n * (head_dim / 16);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define SPATTEN_APPEND(T)                                                                                          \
  hipLaunchKernelGGL((kv_append_kernel<T>), grid, dim3(256), 0, st, (const T*)k_new, (const T*)v_new, new_sb, new_sh, \
                     new_sn, (T*)k_cache, (T*)kr_cache, (T*)v_cache, kv_sb, kv_sh, (const T*)cos, (const T*)sin,       \
                     table_rows, batch, kv_heads, n, head_dim, row0)
  switch (dtype) {
    case SPATTEN_F32: SPATTEN_APPEND(float); break;
    case SPATTEN_F16: SPATTEN_APPEND(f16_t); break;
    default: SPATTEN_APPEND(bf16_t);
  }
#undef SPATTEN_APPEND
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

// C++-linkage name for the other translation units (prefill_attn.hip rotates queries into a workspace with it)
int rope_rows(int dtype, const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sn, void* y, int64_t y_sb, int64_t y_sh,
              int64_t y_sn, const void* cos, const void* sin, int table_rows, const int64_t* position_ids, int64_t pos_sb,
              int pos0, int batch, int heads, int n, int head_dim, void* stream) {
  return spatten_rope_single(dtype, x, x_sb, x_sh, x_sn, y, y_sb, y_sh, y_sn, cos, sin, table_rows, position_ids, pos_sb, pos0,
                             batch, heads, n, head_dim, stream);
}

extern "C" int spatten_abi_version(void) { return SPATTEN_ABI_VERSION; }

extern "C" const char* spatten_status_string(int status) {
  switch (status) {
    case SPATTEN_OK: return "ok";
    case SPATTEN_ERR_INVALID: return "invalid argument";
    case SPATTEN_ERR_UNSUPPORTED: return "unsupported shape";
    case SPATTEN_ERR_WINDOW: return "top-k window holds fewer than k candidates";
    case SPATTEN_ERR_LAUNCH: return "kernel launch failed";
    case SPATTEN_ERR_TIMEOUT: return "a kernel gave up waiting for another workgroup's data";
    default: return "unknown status";
  }
}
