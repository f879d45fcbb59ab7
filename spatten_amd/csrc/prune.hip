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

// The same selection with the whole window in REGISTERS (windows up to 32768 scores): one workgroup of NT threads per
// (layer, head), VPT consecutive scores per thread, read once.  The k-th largest key is found two bits at a time: per
// pass every key is classed 0..3 against the prefix found so far (saturating subtract, shift, min) and counted in a
// byte field of one register — five VALU operations per key, no shuffles and no same-address LDS atomics (real score
// windows concentrate in a handful of exponent bins, which serialises a histogram) — then one DPP wave reduction and
// two LDS adds per wave.  The compaction is ONE packed (greater, equal) block scan.  Same result as
// topk_select_kernel bit for bit (same keys, same tie rule).
template <int CTRL>
__device__ inline unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
__device__ inline unsigned wave_sum_u32(unsigned v) {
  v += dpp_u32<kDppXor1>(v);
  v += dpp_u32<kDppXor2>(v);
  v += dpp_u32<kDppHalfMirror>(v);
  v += dpp_u32<kDppMirror>(v);
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = r[0] + r[1];
  r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return r[0] + r[1];
}

template <typename T, int NT, int VPT>
__global__ __launch_bounds__(NT) void topk_select_reg_kernel(const SelectParams<T> p) {
  constexpr int NW = NT / kWave;
  constexpr int NE = VPT * NW;                 // (slab, wave) cells of 64 consecutive scores, in window order
  constexpr int EPL = (NE + kWave - 1) / kWave;
  __shared__ unsigned s_tot[3][2];
  __shared__ unsigned s_cell[NE + 1];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int h = blockIdx.x % p.H, layer = blockIdx.x / p.H;
  const T* row = (p.score_ptrs ? (const T*)p.score_ptrs[layer] : p.score) + h * p.score_sh + p.lo;
  int32_t* out = p.idx + layer * p.idx_sl + h * p.idx_sh;
  const int W = p.hi - p.lo;
  constexpr int kBits = sizeof(T) == 4 ? 32 : (DT<T>::kId == SPATTEN_BF16 ? 16 : 24);
  constexpr unsigned kKeyMask = kBits == 32 ? 0xFFFFFFFFu : (kBits == 24 ? 0xFFFFFF00u : 0xFFFF0000u);
  const unsigned k = (unsigned)p.k;

  // score j * NT + tid in register j: every load instruction of a wave reads 64 consecutive scores (r02: consecutive
  // scores per THREAD cost 7 us of address-divergent 2-byte loads at 16384 scores)
  unsigned key[VPT];
  {
    float f[VPT];                          // unconditional (clamped) loads, all VPT in flight at once: left to itself
#pragma unroll                             // the compiler sinks each load into its `i < W` branch, one round trip each
    for (int j = 0; j < VPT; ++j) f[j] = DT<T>::to_f32(row[min(j * NT + tid, W - 1)]);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      asm volatile("" : "+v"(f[j]));
      key[j] = j * NT + tid < W ? (ordered_key(f[j]) & kKeyMask) : 0u;   // 0: below every real key
    }
  }
  const int jmax = (W + NT - 1) / NT;    // registers j >= jmax are padding in every thread: skipped below
  if (tid < 6) (&s_tot[0][0])[tid] = 0u;
  __syncthreads();
#if defined(SPATTEN_SEL_EXP) && SPATTEN_SEL_EXP == 1
  { unsigned x = 0; for (int j = 0; j < VPT; ++j) x ^= key[j]; if (x == 0x12345u) out[0] = 1; return; }
#endif

  unsigned t = 0;
  int pass = 0;
#pragma unroll 1
  for (int s = 30; s >= 32 - kBits; s -= 2, ++pass) {
    const int slot = pass % 3;
    unsigned acc = 0;                  // four byte counters: how many of my keys fall in class 0..3
    if (jmax == VPT) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const unsigned x = __builtin_elementwise_sub_sat(key[j], t);
        const unsigned d = min(x >> s, 3u);
        acc += 1u << (8u * d);
      }
    } else {                           // a window that fills only part of the registers: the all-padding ones are skipped
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        if (j >= jmax) break;
        const unsigned x = __builtin_elementwise_sub_sat(key[j], t);
        const unsigned d = min(x >> s, 3u);
        acc += 1u << (8u * d);
      }
    }
    const unsigned f3 = acc >> 24, f2 = (acc >> 16) & 255u, f1 = (acc >> 8) & 255u;
    const unsigned ge3 = f3, ge2 = f3 + f2, ge1 = ge2 + f1;
    const unsigned a = wave_sum_u32(ge1 | (ge2 << 16)), b3 = wave_sum_u32(ge3);
    if (lane == 0) { atomicAdd(&s_tot[slot][0], a); atomicAdd(&s_tot[slot][1], b3); }
    if (tid < 2) s_tot[(pass + 1) % 3][tid] = 0u;     // last read two barriers ago
    __syncthreads();
    const unsigned t0 = s_tot[slot][0], n3 = s_tot[slot][1], n1 = t0 & 0xFFFFu, n2 = t0 >> 16;
    t |= (n3 >= k ? 3u : (n2 >= k ? 2u : (n1 >= k ? 1u : 0u))) << s;
  }
#if defined(SPATTEN_SEL_EXP) && SPATTEN_SEL_EXP == 2
  { if (t == 0x12345u) out[0] = 1; return; }
#endif
  // t = key of the k-th largest.  Order-preserving compaction: packed (greater, equal) counts per cell, one scan over
  // the cells, lane offsets from the ballots.
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const unsigned long long m_gt = __ballot(key[j] > t);
    const unsigned long long m_eq = __ballot(key[j] == t && j * NT + tid < W);
    if (lane == 0) s_cell[j * NW + wave] = ((unsigned)__popcll(m_gt) << 16) | (unsigned)__popcll(m_eq);   // <= 32768 each
  }
  __syncthreads();
  if (wave == 0) {
    unsigned c[EPL], tot = 0;
#pragma unroll
    for (int u = 0; u < EPL; ++u) { c[u] = (lane * EPL + u < NE) ? s_cell[lane * EPL + u] : 0u; tot += c[u]; }
    unsigned inc = tot;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned v = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += v;
    }
    unsigned ex = inc - tot;
#pragma unroll
    for (int u = 0; u < EPL; ++u) { if (lane * EPL + u < NE) s_cell[lane * EPL + u] = ex; ex += c[u]; }
    if (lane == kWave - 1) s_cell[NE] = inc;
  }
  __syncthreads();
  const unsigned need_eq = k - (s_cell[NE] >> 16);      // >= 1: that many keys equal to t are kept, lowest index first
#if defined(SPATTEN_SEL_EXP) && SPATTEN_SEL_EXP == 3
  { if (need_eq == 0x12345u) out[0] = 1; return; }
#endif
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int i = j * NT + tid;
    const bool gt = key[j] > t, eq = key[j] == t && i < W;
    const unsigned long long m_gt = __ballot(gt), m_eq = __ballot(eq);
    const unsigned ex = s_cell[j * NW + wave];
    const unsigned g = (ex >> 16) + __builtin_amdgcn_mbcnt_hi((unsigned)(m_gt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_gt, 0u));
    const unsigned e = (ex & 0xFFFFu) + __builtin_amdgcn_mbcnt_hi((unsigned)(m_eq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_eq, 0u));
    if (gt) out[g + (e < need_eq ? e : need_eq)] = p.lo + i;
    else if (eq && e < need_eq) out[g + e] = p.lo + i;
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
  static int env_fast = -1;
  if (env_fast < 0) { const char* e = getenv("SPATTEN_SELECT_REG"); env_fast = e ? atoi(e) : 1; }
  const int W = hi - lo;
  const dim3 grid((unsigned)(layers * H));
  // Many windows (the prune event: layers x heads of them): 4-wave workgroups, several resident per CU, cheap barriers.
  // Few long windows (local V pruning: one per head): 16 waves on one window.
  const bool many = layers * H >= 128;
#define SPATTEN_SEL(NT_, VPT_) hipLaunchKernelGGL((topk_select_reg_kernel<T, NT_, VPT_>), grid, dim3(NT_), 0, st, p)
  if (!env_fast || W > 32768) hipLaunchKernelGGL((topk_select_kernel<T>), grid, dim3(kSelThreads), 0, st, p);
  else if (W <= 1024) SPATTEN_SEL(256, 4);
  else if (many && W <= 4096) SPATTEN_SEL(256, 16);
  else if (many && W <= 8192) SPATTEN_SEL(256, 32);
  else if (W <= 4096) SPATTEN_SEL(1024, 4);
  else if (W <= 16384) SPATTEN_SEL(1024, 16);
  else SPATTEN_SEL(1024, 32);
#undef SPATTEN_SEL
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

// all layers' fp32 accumulator rows through the prune's row map (start | idx | tail): blockIdx.z = layer
__global__ __launch_bounds__(256) void acc_compact_layers_kernel(const float* const* __restrict__ src_ptrs, int64_t src_sh,
                                                                 float* const* __restrict__ dst_ptrs, int64_t dst_sh,
                                                                 const int32_t* __restrict__ idx, int64_t idx_sl,
                                                                 int64_t idx_sh, int start, int k, int tail_lo, int Lp) {
  const int r = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, l = blockIdx.z;
  if (r >= Lp) return;
  int j;
  if (r < start) j = r;
  else if (r < start + k) j = idx[l * idx_sl + h * idx_sh + (r - start)];
  else j = tail_lo + (r - start - k);
  dst_ptrs[l][h * dst_sh + r] = src_ptrs[l][h * src_sh + j];
}

extern "C" int spatten_prune_layers_scored(int score_dtype, int kv_dtype, int layers, const void* const* score_ptrs,
                                           int64_t score_sh, const void* const* k_src_ptrs, const void* const* v_src_ptrs,
                                           int64_t src_sb, int64_t src_sh, void* const* k_dst_ptrs, void* const* v_dst_ptrs,
                                           void* const* kr_dst_ptrs, int64_t dst_sb, int64_t dst_sh, const void* cos,
                                           const void* sin, int table_rows, int32_t* idx,
                                           const float* const* acc_src_ptrs, int64_t acc_src_sh, float* const* acc_dst_ptrs,
                                           int64_t acc_dst_sh, int batch, int heads, int head_dim, int lo, int hi, int k,
                                           int tail_lo, int tail_len, void* stream) {
  if (!score_ptrs || !k_src_ptrs || !v_src_ptrs || !k_dst_ptrs || !v_dst_ptrs || !idx) return SPATTEN_ERR_INVALID;
  if ((acc_src_ptrs == nullptr) != (acc_dst_ptrs == nullptr)) return SPATTEN_ERR_INVALID;
  const int rc = select_any(score_dtype, nullptr, score_ptrs, score_sh, layers, heads, lo, hi, k, idx,
                            (int64_t)heads * k, k, (hipStream_t)stream);
  if (rc != SPATTEN_OK) return rc;
  CompactParams p{};
  p.k_src_ptrs = k_src_ptrs; p.v_src_ptrs = v_src_ptrs; p.k_dst_ptrs = k_dst_ptrs; p.v_dst_ptrs = v_dst_ptrs;
  p.kr_dst_ptrs = kr_dst_ptrs; p.cos = cos; p.sin = sin; p.table_rows = table_rows;
  p.src_sb = src_sb; p.src_sh = src_sh; p.dst_sb = dst_sb; p.dst_sh = dst_sh;
  p.idx = idx; p.idx_sl = (int64_t)heads * k; p.idx_sh = k;
  p.B = batch; p.H = heads; p.layers = layers; p.n_tensors = 2;
  p.start = lo; p.k = k; p.tail_lo = tail_lo;
  const int rc2 = compact_any(kv_dtype, p, head_dim, tail_len, (hipStream_t)stream);
  if (rc2 != SPATTEN_OK || !acc_src_ptrs) return rc2;
  const int Lp = lo + k + tail_len;
  if (heads > 65535 || layers > 65535) return SPATTEN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(acc_compact_layers_kernel, dim3((unsigned)ceil_div(Lp, 256), (unsigned)heads, (unsigned)layers), dim3(256),
                     0, (hipStream_t)stream, acc_src_ptrs, acc_src_sh, acc_dst_ptrs, acc_dst_sh, idx, (int64_t)heads * k,
                     (int64_t)k, lo, k, tail_lo, Lp);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
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
