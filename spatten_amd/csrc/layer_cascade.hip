// layer_cascade.hip — the prune event of the layer-to-layer cascade (README.md:11; the traces' `key_fetch_num` shrinks
// layer by layer: spatten_hardware/hardware/workloads/*.csv) for ALL layers in three launches.
// PARITY UNPINNED (no numeric implementation in the reference; oracle: layer_cascade_prune).
//
// Layer l keeps k_l tokens of its window, ranked by its own importance but among the tokens layer l-1 just kept.  That
// dependency runs PER HEAD — head h of layer l only needs head h's kept ids of layer l-1 — so one workgroup per head
// walks the layers in order and no launch boundary separates them (round 2: a Python loop of rank -> select -> id gather
// -> K/V gather per layer, ~10 launches x 32 layers).  Then ONE ragged gather moves K / V / shadow rows of every layer
// (the layers keep different numbers of rows, so lengths and strides come from a per-layer table), and one moves the
// cascade accumulators.
#include <algorithm>
#include <mutex>

#include "common.h"

#ifdef SPATTEN_LC_TRACE   // developer instrumentation: phase timestamps of head 0's chain, 8 slots per layer (tools/mb/lc_trace.py)
__device__ unsigned long long* g_lc_trace = nullptr;
#define LC_STAMP(layer, slot)                                                                         \
  do { if (g_lc_trace && blockIdx.x == 0 && threadIdx.x == 0) g_lc_trace[(layer) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define LC_STAMP(layer, slot)
#endif

namespace spatten {

// per-layer geometry, DEVICE array (all lengths in rows / elements)
struct LayerPrune {
  int64_t len, hi, k, new_len;         // cache length, window end (= tail start), kept in the window, new length
  int64_t score_sh, n_known, known_sh, new_ids_sh;   // importance row stride; token ids known for slots [0, n_known)
  int64_t src_sb, src_sh, dst_sb, dst_sh;            // K / V planes (elements)
  int64_t acc_src_sh, acc_dst_sh, id_base, pad_;     // fp32 accumulators; ids of slots >= n_known are id_base + (j - n_known)
};
static_assert(sizeof(LayerPrune) == 128, "table stride");

struct ChainParams {
  const LayerPrune* lay;
  const void* const* score_ptrs;       // [layers] -> [H, >= len] scores (model dtype or fp32)
  const int32_t* const* known_ptrs;    // [layers] -> int32 [H, n_known] token ids of the slots seen by the last prune (may be NULL rows)
  int32_t* const* new_ids_ptrs;        // [layers] -> int32 [H, new_len] out: ids of the new cache's slots
  int32_t* idx; int64_t idx_sl, idx_sh;   // [layers, H, kmax] out: kept window positions, ascending
  uint32_t* keys; int64_t keys_sh;     // [H, >= max window] scratch (windows that do not fit the LDS copy)
  int layers, start, lds_keys, heads;  // windows up to lds_keys entries keep their keys in LDS
  int l_begin, l_end;                  // this launch walks layers [l_begin, l_end) (the event = a few launches: see the host side)
  int map_ids;                         // ... and behind them a byte per token id in [map_base, map_base + map_ids): the membership map
};

__device__ inline int gridDim_heads(const ChainParams& p) { return p.heads; }
constexpr int kChainThreads = 1024;    // the chain is latency-bound on ONE workgroup per head: many threads, few iterations
constexpr int kChainWaves = kChainThreads / 64;
constexpr int kChainMaxLayers = 48;    // layers per launch whose tables live in LDS (the host cuts longer events into more legs)

// the chain's barrier: LDS traffic only.  __syncthreads() carries a fence = s_waitcnt vmcnt(0), which would wait for the NEXT
// layer's prefetch (a global round trip per layer: r04 stamps, 6k of 26k cycles) — a bare s_barrier behind the LDS counter does not.
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ inline int32_t slot_id(const LayerPrune& L, const int32_t* known, int j) {
  return j < (int)L.n_known ? known[j] : (int32_t)(L.id_base + (j - L.n_known));
}

// one workgroup per head walks the layers: rank (membership in the previous layer's kept ids -> score or -inf), exact top-k
// of the window (radix select, ties: lowest position), the new slot ids.  Same keys, tie rule and output order as
// cascade_rank_kernel + topk_select_kernel<float> + the id gather of round 2.
// PUBLISH: the kept positions go out write-through and a per-(layer, head) word is raised when they are complete.  (Round 4
// built the event as ONE launch on it — 32 chain workgroups + 224 persistent gather workers waiting on those words, the
// gather of layer l under the chain's layer l + 1 — bit-identical, and SLOWER: 1,061 us against 970 for the launches below at
// Llama-2-7B geometry: the chain's LDS / barrier path does not get shorter, its global accesses get slower under the
// gather's 5 TB/s, and 224 one-item-at-a-time workers gather more slowly than the free-running grid.  Removed again; the
// flag is kept for the record and instantiated false.)
template <typename T, bool PUBLISH>
__device__ __forceinline__ void chain_body(const ChainParams& p, const int h, unsigned* ready, const unsigned gen) {
  // kHistCopies private histograms (wave w counts into copy w % kHistCopies): real score windows fall into a handful of
  // exponent bins, and same-address LDS atomics serialise (r03: one shared histogram was most of the chain's time)
  constexpr int kHistCopies = 8;
  __shared__ unsigned s_hist[kHistCopies][256];
  __shared__ unsigned s_sel[2];
  __shared__ unsigned s_cnt[2][kChainWaves][2];
  extern __shared__ uint32_t s_keys[];
  // keys of model-dtype scores populate the top 16 (bf16) / 19 (f16) bits; the bits below only repeat the sign (all ones for
  // negative values, -inf of a non-member included), so the digits below them cannot separate two keys: skipped
  constexpr int kPasses = sizeof(T) == 4 ? 4 : (DT<T>::kId == SPATTEN_BF16 ? 2 : 3);
  constexpr unsigned kKeyMask = kPasses == 4 ? 0xFFFFFFFFu : (kPasses == 3 ? 0xFFFFFF00u : 0xFFFF0000u);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Membership of a slot's token in the previous layer's new cache.  r03 kept that layer's id row in LDS and binary-searched
  // it: 11-12 DEPENDENT LDS loads per key, 3 keys per thread — 36 % of the chain (tools/mb/lc_trace.py: 11.3k of 31.4k cycles per
  // layer).  Round 4: a byte per token id, s_map[id - base] = l + 1 once layer l keeps the token — written by the thread that
  // writes the id row anyway, never cleared (the stamp of layer l - 1 is what layer l asks for), one LDS load per key.  Ids
  // outside [base, base + map_ids) (a conversation of more than ~98k tokens) fall back to the binary search of the global row.
  // the layers' tables and row pointers: copied into LDS once (a layer start was two dependent scalar-load round trips to global memory)
  __shared__ LayerPrune s_lay[kChainMaxLayers];
  __shared__ const void* s_score[kChainMaxLayers];
  __shared__ const int32_t* s_known[kChainMaxLayers];
  __shared__ int32_t* s_nid[kChainMaxLayers];
  for (int i = tid; i < (p.l_end - p.l_begin) * (int)(sizeof(LayerPrune) / 8); i += kChainThreads)
    reinterpret_cast<uint64_t*>(s_lay)[i] = reinterpret_cast<const uint64_t*>(p.lay + p.l_begin)[i];
  if (tid < p.l_end - p.l_begin) {
    s_score[tid] = p.score_ptrs[p.l_begin + tid];
    s_known[tid] = p.known_ptrs[p.l_begin + tid];
    s_nid[tid] = p.new_ids_ptrs[p.l_begin + tid];
  }
  uint8_t* s_map = reinterpret_cast<uint8_t*>(s_keys + p.lds_keys);
  const int R = p.map_ids;
  for (int i = tid * 4; i < R; i += kChainThreads * 4) *reinterpret_cast<uint32_t*>(s_map + i) = 0u;
  int32_t map_base;
  {
    const LayerPrune L0 = p.lay[p.l_begin];
    const int32_t* known0 = p.known_ptrs[p.l_begin] ? p.known_ptrs[p.l_begin] + h * L0.known_sh : nullptr;
    map_base = slot_id(L0, known0, p.start);            // the oldest prunable token of this head (ids ascend along the slots)
  }
  auto stamp = [&](int32_t id, int l) {
    const unsigned off = (unsigned)(id - map_base);
    if (off < (unsigned)R) s_map[off] = (uint8_t)(l + 1);
  };
  const int32_t* prev = nullptr;       // the previous layer's id row in global memory (fallback; null: no previous layer)
  int n_prev = 0;
  // Everything a layer reads from global memory — its scores, the token ids of its slots — does NOT depend on what the
  // previous layer selected, so it is requested ONE LAYER AHEAD into registers (round 4): the per-layer critical path is then
  // LDS traffic and barriers only (r03: 3-4 dependent global round trips per layer, 12 us per layer on one workgroup; under
  // the overlapped event's gather traffic those round trips are 2-3x longer).  Thread t holds window positions t + 1024 q,
  // q < kPf, the tail rows t + 1024 q, q < kPt, and the head row t; longer windows / tails load the rest in place.
  constexpr int kPf = 4, kPt = 2;
  T pf_sc[kPf];
  int32_t pf_id[kPf], pf_tail[kPt], pf_head = 0;
  auto prefetch = [&](int l, bool tables_in_lds) {
    const LayerPrune L = tables_in_lds ? s_lay[l - p.l_begin] : p.lay[l];
    const void* sp = tables_in_lds ? s_score[l - p.l_begin] : p.score_ptrs[l];
    const int32_t* kp = tables_in_lds ? s_known[l - p.l_begin] : p.known_ptrs[l];
    const T* score = (const T*)sp + h * L.score_sh;
    const int32_t* known = kp ? kp + h * L.known_sh : nullptr;
    const int W = (int)L.hi - p.start, tail = (int)(L.len - L.hi);
#pragma unroll
    for (int q = 0; q < kPf; ++q) {
      const int i = min(tid + q * kChainThreads, max(W - 1, 0)), j = p.start + i;
      pf_sc[q] = score[j];
      pf_id[q] = slot_id(L, known, j);
    }
#pragma unroll
    for (int q = 0; q < kPt; ++q) pf_tail[q] = slot_id(L, known, (int)L.hi + min(tid + q * kChainThreads, max(tail - 1, 0)));
    pf_head = slot_id(L, known, min(tid, max(p.start - 1, 0)));
  };
  prefetch(p.l_begin, false);
  lds_sync();
  if (p.l_begin > 0) {      // a later leg of the chain: the previous layer's id row was written by the launch before this one
    const LayerPrune Lp = p.lay[p.l_begin - 1];
    prev = p.new_ids_ptrs[p.l_begin - 1] + h * Lp.new_ids_sh;
    n_prev = (int)Lp.new_len;
    for (int i = tid; i < n_prev; i += kChainThreads) stamp(prev[i], p.l_begin - 1);
    lds_sync();
  }
  for (int l = p.l_begin; l < p.l_end; ++l) {
    const LayerPrune L = s_lay[l - p.l_begin];
    const T* score = (const T*)s_score[l - p.l_begin] + h * L.score_sh;
    const int32_t* known = s_known[l - p.l_begin] ? s_known[l - p.l_begin] + h * L.known_sh : nullptr;
    const int W = (int)L.hi - p.start, k = (int)L.k;
    int32_t* out = p.idx + l * p.idx_sl + h * p.idx_sh;
    uint32_t* keys = W <= p.lds_keys ? s_keys : p.keys + h * p.keys_sh;
    // this layer's prefetched values move to their own registers: the prefetch of layer l + 1 reuses pf_*
    T my_sc[kPf];
    int32_t my_id[kPf], my_tail[kPt];
    const int32_t my_head = pf_head;
#pragma unroll
    for (int q = 0; q < kPf; ++q) { my_sc[q] = pf_sc[q]; my_id[q] = pf_id[q]; }
#pragma unroll
    for (int q = 0; q < kPt; ++q) my_tail[q] = pf_tail[q];
    LC_STAMP(l, 0);
    if (l + 1 < p.l_end) prefetch(l + 1, true);
    // ---- keys of the window
    {
      int q = 0;
      for (int i = tid; i < W; i += kChainThreads, ++q) {
        const int j = p.start + i;
        bool member = true;
        T scv = my_sc[0];
        int32_t want = 0;
        if (q < kPf) {
#pragma unroll
          for (int qq = 0; qq < kPf; ++qq) if (qq == q) { scv = my_sc[qq]; want = my_id[qq]; }
        } else {
          scv = score[j];
          want = slot_id(L, known, j);
        }
        if (prev) {
          const unsigned off = (unsigned)(want - map_base);
          if (off < (unsigned)R) {
            member = s_map[off] == (uint8_t)l;           // the stamp of layer l - 1
          } else {
            int lo = 0, hi = n_prev;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (prev[mid] < want) lo = mid + 1; else hi = mid; }
            member = lo < n_prev && prev[lo] == want;
          }
        }
        keys[i] = ordered_key(member ? DT<T>::to_f32(scv) : -INFINITY) & kKeyMask;
      }
    }
    lds_sync();
    LC_STAMP(l, 1);
    // ---- the key of the k-th largest (8-bit digits, most significant first)
    unsigned prefix = 0, pmask = 0, k_rem = (unsigned)k;
#pragma unroll 1
    for (int pass = 0; pass < kPasses; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int t = tid; t < kHistCopies * 256; t += kChainThreads) (&s_hist[0][0])[t] = 0;
      lds_sync();
      unsigned* my_hist = s_hist[wave % kHistCopies];
      for (int i = tid; i < W; i += kChainThreads) {
        const unsigned key = keys[i];
        if ((key & pmask) == prefix) atomicAdd(&my_hist[(key >> shift) & 255u], 1u);
      }
      lds_sync();
      if (wave == 0) {
        unsigned c[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c[j] = 0;
#pragma unroll
          for (int q = 0; q < kHistCopies; ++q) c[j] += s_hist[q][255 - 4 * lane - j];
          tot += c[j];
        }
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
      lds_sync();
      prefix |= s_sel[0] << shift;
      pmask |= 255u << shift;
      k_rem -= s_sel[1];
    }
    const unsigned thr = prefix, need_eq = k_rem;
    LC_STAMP(l, 2);
    // ---- order-preserving compaction: everything above the threshold, the first need_eq at it.  The thread that keeps
    // window position i also knows that slot's token id (prefetched): the new cache's id row is written here, not gathered
    int32_t* nid = s_nid[l - p.l_begin] + h * L.new_ids_sh;
    const int lp = (int)L.new_len;
    unsigned run_eq = 0, run_kept = 0;
    int parity = 0, q = 0;
    for (int base = 0; base < W; base += kChainThreads, parity ^= 1, ++q) {
      const int i = base + tid;
      const bool in = i < W;
      const unsigned key = in ? keys[i] : 0u;
      const bool gt = in && key > thr, eq = in && key == thr;
      const unsigned long long m_gt = __ballot(gt), m_eq = __ballot(eq);
      if (lane == 0) { s_cnt[parity][wave][0] = __popcll(m_gt); s_cnt[parity][wave][1] = __popcll(m_eq); }
      lds_sync();
      unsigned eq_base = run_eq, kept_base = run_kept, tot_eq = run_eq, tot_kept = run_kept;
#pragma unroll
      for (int w = 0; w < kChainWaves; ++w) {
        const unsigned g = s_cnt[parity][w][0], e = s_cnt[parity][w][1];
        const unsigned room = tot_eq < need_eq ? need_eq - tot_eq : 0u;
        const unsigned kept_w = g + (e < room ? e : room);
        if (w < wave) { eq_base += e; kept_base += kept_w; }
        tot_eq += e;
        tot_kept += kept_w;
      }
      const unsigned long long lt = (1ull << lane) - 1ull;
      const bool keep = gt || (eq && eq_base + __popcll(m_eq & lt) < need_eq);
      const unsigned pos = kept_base + __popcll(__ballot(keep) & lt);
      if (keep && pos < (unsigned)k) {
        if (PUBLISH) __hip_atomic_store(out + pos, p.start + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
        else out[pos] = p.start + i;
        int32_t id = 0;
        if (q < kPf) {
#pragma unroll
          for (int qq = 0; qq < kPf; ++qq) if (qq == q) id = my_id[qq];
        } else {
          id = slot_id(L, known, p.start + i);
        }
        nid[p.start + pos] = id;
        stamp(id, l);
      }
      run_eq = tot_eq;
      run_kept = tot_kept;
    }
    LC_STAMP(l, 3);
    // ---- the ids of the head rows [0, start) and of the tail rows [hi, len) of the new cache (prefetched)
    if (tid < p.start) { nid[tid] = my_head; stamp(my_head, l); }
    for (int r = kChainThreads; r + tid < p.start; r += kChainThreads) {      // (start > 1024: not prefetched)
      const int32_t id = slot_id(L, known, r + tid);
      nid[r + tid] = id; stamp(id, l);
    }
    {
      const int tail = (int)(L.len - L.hi);
      int qt = 0;
      for (int t = tid; t < tail; t += kChainThreads, ++qt) {
        int32_t id = 0;
        if (qt < kPt) {
#pragma unroll
          for (int qq = 0; qq < kPt; ++qq) if (qq == qt) id = my_tail[qq];
        } else {
          id = slot_id(L, known, (int)L.hi + t);
        }
        const int r = p.start + k + t;
        nid[r] = id;
        stamp(id, l);
      }
    }
    if (PUBLISH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its own stores ...
    lds_sync();
    if (PUBLISH && tid == 0)                                            // ... then ONE lane raises the layer's word
      __hip_atomic_store(ready + (size_t)l * gridDim_heads(p) + h, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = nid;
    n_prev = lp;
    LC_STAMP(l, 4);
  }
}

template <typename T>
__global__ __launch_bounds__(kChainThreads) void layer_cascade_select_kernel(const ChainParams p) {
  chain_body<T, false>(p, (int)blockIdx.x, nullptr, 0u);
}

// ---- the BLOCKED chain (round 4): windows of up to 4 x 1024 positions, thread t owns the four CONSECUTIVE positions 4t .. 4t+3.
// The phase stamps of chain_body (profiles/r04_layer_cascade_anatomy.txt) are barriers and LDS round trips: 11 workgroup barriers
// of 16 waves per layer — three per radix pass (clear, count, wave 0's scan), three chunk rounds of the compaction.  Here the keys
// stay in registers, every radix pass has its own histogram (cleared while the previous layer's rows are written), EVERY wave
// scans the histogram itself (no broadcast round), and with a blocked layout the order-preserving compaction is one prefix over
// threads (wave scan + the 16 wave totals): 4 barriers per layer.  Same keys, tie rule (lowest position first) and outputs as
// chain_body, bit for bit (tests/test_gpu_cascade.py runs both).
constexpr int kBlk = 4;                       // window positions per thread
constexpr int kBlkCopies = 4;                 // private histograms per pass
template <typename T>
__global__ __launch_bounds__(kChainThreads) void layer_cascade_select_blocked_kernel(const ChainParams p) {
  constexpr int kPasses = sizeof(T) == 4 ? 4 : (DT<T>::kId == SPATTEN_BF16 ? 2 : 3);
  constexpr unsigned kKeyMask = kPasses == 4 ? 0xFFFFFFFFu : (kPasses == 3 ? 0xFFFFFF00u : 0xFFFF0000u);
  __shared__ unsigned s_hist[kPasses][kBlkCopies][256];
  __shared__ unsigned s_tot[kChainWaves][2];
  __shared__ LayerPrune s_lay[kChainMaxLayers];
  __shared__ const void* s_score[kChainMaxLayers];
  __shared__ const int32_t* s_known[kChainMaxLayers];
  __shared__ int32_t* s_nid[kChainMaxLayers];
  extern __shared__ uint32_t s_dyn[];
  uint8_t* s_map = reinterpret_cast<uint8_t*>(s_dyn);
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = p.map_ids;
  const int nl = p.l_end - p.l_begin;
  for (int i = tid; i < nl * (int)(sizeof(LayerPrune) / 8); i += kChainThreads)
    reinterpret_cast<uint64_t*>(s_lay)[i] = reinterpret_cast<const uint64_t*>(p.lay + p.l_begin)[i];
  if (tid < nl) {
    s_score[tid] = p.score_ptrs[p.l_begin + tid];
    s_known[tid] = p.known_ptrs[p.l_begin + tid];
    s_nid[tid] = p.new_ids_ptrs[p.l_begin + tid];
  }
  for (int i = tid * 4; i < R; i += kChainThreads * 4) *reinterpret_cast<uint32_t*>(s_map + i) = 0u;
  for (int i = tid; i < kPasses * kBlkCopies * 256; i += kChainThreads) (&s_hist[0][0][0])[i] = 0u;
  int32_t map_base;
  {
    const LayerPrune L0 = p.lay[p.l_begin];
    const int32_t* known0 = p.known_ptrs[p.l_begin] ? p.known_ptrs[p.l_begin] + h * L0.known_sh : nullptr;
    map_base = slot_id(L0, known0, p.start);
  }
  auto stamp = [&](int32_t id, int l) {
    const unsigned off = (unsigned)(id - map_base);
    if (off < (unsigned)R) s_map[off] = (uint8_t)(l + 1);
  };
  const int32_t* prev = nullptr;
  int n_prev = 0;
  constexpr int kPt = 2;
  T pf_sc[kBlk];
  int32_t pf_id[kBlk], pf_tail[kPt], pf_head = 0;
  auto prefetch = [&](int l, bool tables_in_lds) {
    const LayerPrune L = tables_in_lds ? s_lay[l - p.l_begin] : p.lay[l];
    const void* sp = tables_in_lds ? s_score[l - p.l_begin] : p.score_ptrs[l];
    const int32_t* kp = tables_in_lds ? s_known[l - p.l_begin] : p.known_ptrs[l];
    const T* score = (const T*)sp + h * L.score_sh;
    const int32_t* known = kp ? kp + h * L.known_sh : nullptr;
    const int W = (int)L.hi - p.start, tail = (int)(L.len - L.hi);
#pragma unroll
    for (int q = 0; q < kBlk; ++q) {
      const int i = min(kBlk * tid + q, max(W - 1, 0)), j = p.start + i;
      pf_sc[q] = score[j];
      pf_id[q] = slot_id(L, known, j);
    }
#pragma unroll
    for (int q = 0; q < kPt; ++q) pf_tail[q] = slot_id(L, known, (int)L.hi + min(tid + q * kChainThreads, max(tail - 1, 0)));
    pf_head = slot_id(L, known, min(tid, max(p.start - 1, 0)));
  };
  prefetch(p.l_begin, false);
  lds_sync();
  if (p.l_begin > 0) {
    const LayerPrune Lp = p.lay[p.l_begin - 1];
    prev = p.new_ids_ptrs[p.l_begin - 1] + h * Lp.new_ids_sh;
    n_prev = (int)Lp.new_len;
    for (int i = tid; i < n_prev; i += kChainThreads) stamp(prev[i], p.l_begin - 1);
    lds_sync();
  }
  for (int l = p.l_begin; l < p.l_end; ++l) {
    const LayerPrune L = s_lay[l - p.l_begin];
    const int32_t* known = s_known[l - p.l_begin] ? s_known[l - p.l_begin] + h * L.known_sh : nullptr;
    const int W = (int)L.hi - p.start, k = (int)L.k;
    int32_t* out = p.idx + l * p.idx_sl + h * p.idx_sh;
    T my_sc[kBlk];
    int32_t my_id[kBlk], my_tail[kPt];
    const int32_t my_head = pf_head;
#pragma unroll
    for (int q = 0; q < kBlk; ++q) { my_sc[q] = pf_sc[q]; my_id[q] = pf_id[q]; }
#pragma unroll
    for (int q = 0; q < kPt; ++q) my_tail[q] = pf_tail[q];
    LC_STAMP(l, 0);
    if (l + 1 < p.l_end) prefetch(l + 1, true);
    // ---- keys (registers)
    unsigned key[kBlk];
    bool in[kBlk];
#pragma unroll
    for (int q = 0; q < kBlk; ++q) {
      in[q] = kBlk * tid + q < W;
      bool member = true;
      if (prev) {
        const unsigned off = (unsigned)(my_id[q] - map_base);
        if (off < (unsigned)R) {
          member = s_map[off] == (uint8_t)l;
        } else {
          int lo = 0, hi = n_prev;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (prev[mid] < my_id[q]) lo = mid + 1; else hi = mid; }
          member = lo < n_prev && prev[lo] == my_id[q];
        }
      }
      key[q] = ordered_key(member ? DT<T>::to_f32(my_sc[q]) : -INFINITY) & kKeyMask;
    }
    LC_STAMP(l, 1);
    // ---- the key of the k-th largest: one histogram per pass (cleared one layer ago), one barrier per pass, every wave scans
    unsigned prefix = 0, pmask = 0, k_rem = (unsigned)k;
#pragma unroll
    for (int pass = 0; pass < kPasses; ++pass) {
      const int shift = 24 - 8 * pass;
      unsigned* my_hist = s_hist[pass][wave % kBlkCopies];
      // (r04 A/B: aggregating equal digits inside the wave first — ballot + one lane adds the count — is SLOWER on Gaussian scores,
      //  10.8k against 9.2k cycles for the two passes: ~14 distinct exponent digits per wave cost more than the conflicts they save)
#pragma unroll
      for (int q = 0; q < kBlk; ++q)
        if (in[q] && (key[q] & pmask) == prefix) atomicAdd(&my_hist[(key[q] >> shift) & 255u], 1u);
      lds_sync();
      unsigned c[4], tot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j] = 0;
#pragma unroll
        for (int q = 0; q < kBlkCopies; ++q) c[j] += s_hist[pass][q][255 - 4 * lane - j];
        tot += c[j];
      }
      unsigned inc = tot;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned t = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += t;
      }
      unsigned ex = inc - tot, digit = 0, before = 0;
      const bool mine = ex < k_rem && k_rem <= inc;
      if (mine) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ex < k_rem && k_rem <= ex + c[j]) { digit = 255 - 4 * lane - j; before = ex; }
          ex += c[j];
        }
      }
      const unsigned long long hit = __ballot(mine);
      const int src = hit ? (int)__builtin_ctzll(hit) : 0;
      digit = __shfl(digit, src, kWave);
      before = __shfl(before, src, kWave);
      prefix |= digit << shift;
      pmask |= 255u << shift;
      k_rem -= before;
    }
    const unsigned thr = prefix, need_eq = k_rem;
    LC_STAMP(l, 2);
    // ---- order-preserving compaction: prefix over threads of (keys above the threshold, keys at it)
    unsigned n_gt = 0, n_eq = 0;
#pragma unroll
    for (int q = 0; q < kBlk; ++q) { n_gt += (in[q] && key[q] > thr) ? 1u : 0u; n_eq += (in[q] && key[q] == thr) ? 1u : 0u; }
    unsigned packed = n_gt | (n_eq << 16), inc = packed;          // both counts fit 16 bits (<= 4096 positions)
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned t = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += t;
    }
    if (lane == kWave - 1) { s_tot[wave][0] = inc & 0xFFFFu; s_tot[wave][1] = inc >> 16; }
    lds_sync();
    unsigned gt_before = (inc - packed) & 0xFFFFu, eq_before = (inc - packed) >> 16;
#pragma unroll
    for (int w = 0; w < kChainWaves; ++w)
      if (w < wave) { gt_before += s_tot[w][0]; eq_before += s_tot[w][1]; }
    // (the histograms of this layer are dead: every wave scanned them before the barrier above — clear them for the next layer)
    for (int i = tid; i < kPasses * kBlkCopies * 256; i += kChainThreads) (&s_hist[0][0][0])[i] = 0u;
    int32_t* nid = s_nid[l - p.l_begin] + h * L.new_ids_sh;
    {
      unsigned g = gt_before, e = eq_before;
#pragma unroll
      for (int q = 0; q < kBlk; ++q) {
        const bool gt = in[q] && key[q] > thr, eq = in[q] && key[q] == thr;
        const bool keep = gt || (eq && e < need_eq);
        const unsigned pos = g + (e < need_eq ? e : need_eq);
        if (keep) {
          out[pos] = p.start + kBlk * tid + q;
          nid[p.start + pos] = my_id[q];
          stamp(my_id[q], l);
        }
        g += gt ? 1u : 0u;
        e += eq ? 1u : 0u;
      }
    }
    LC_STAMP(l, 3);
    // ---- the ids of the head rows [0, start) and of the tail rows [hi, len) of the new cache
    if (tid < p.start) { nid[tid] = my_head; stamp(my_head, l); }
    for (int r = kChainThreads; r + tid < p.start; r += kChainThreads) {
      const int32_t id = slot_id(L, known, r + tid);
      nid[r + tid] = id; stamp(id, l);
    }
    {
      const int tail = (int)(L.len - L.hi);
      int qt = 0;
      for (int t = tid; t < tail; t += kChainThreads, ++qt) {
        int32_t id = 0;
        if (qt < kPt) {
#pragma unroll
          for (int qq = 0; qq < kPt; ++qq) if (qq == qt) id = my_tail[qq];
        } else {
          id = slot_id(L, known, (int)L.hi + t);
        }
        const int r = p.start + k + t;
        nid[r] = id;
        stamp(id, l);
      }
    }
    lds_sync();
    prev = nid;
    n_prev = (int)L.new_len;
    LC_STAMP(l, 4);
  }
}

struct RaggedParams {
  const LayerPrune* lay;
  const void* const* k_src_ptrs; const void* const* v_src_ptrs;
  void* const* k_dst_ptrs; void* const* v_dst_ptrs; void* const* kr_dst_ptrs;
  const void* cos; const void* sin; int table_rows;
  const int32_t* idx; int64_t idx_sl, idx_sh;
  int B, H, start, row_bytes, es, half_ppr, rows_per_block;
  int l0;                              // first layer of this launch (grid z counts from it)
};

// the fused gather + concat (+ rotated shadow) of kv_compact_kernel with every layer's own lengths and strides
// one 256-thread item: rows [rb * rows_per_block, +rows_per_block) of plane t (0 = K + shadow, 1 = V) of (b, h) of `layer`, in
// two halves so that a caller can put several items' loads in flight before the first store.
// FRESH: the kept positions were published by another workgroup of the SAME launch -> agent-scope loads
struct RaggedRow { u32x4 lo_v, hi_v; int64_t doff; int r; bool live; };
template <typename T, bool FRESH>
__device__ __forceinline__ RaggedRow ragged_load(const RaggedParams& p, const LayerPrune& L, const int layer, const int t, const int bh,
                                                 const int rb, const int tid) {
  RaggedRow o;
  const int rloc = tid / p.half_ppr, piece = tid - rloc * p.half_ppr;
  const int r = rb * p.rows_per_block + rloc;
  o.r = r;
  o.live = rloc < p.rows_per_block && r < (int)L.new_len && rb >= 0;
  o.doff = 0;
  if (!o.live) return o;
  const int b = bh / p.H, h = bh - b * p.H;
  const int k = (int)L.k, half_bytes = p.row_bytes >> 1;
  int src_row;
  if (r < p.start) src_row = r;
  else if (r < p.start + k) {
    const int32_t* ip = p.idx + layer * p.idx_sl + h * p.idx_sh + (r - p.start);
    src_row = FRESH ? __hip_atomic_load(ip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ip;
  } else src_row = (int)L.hi + (r - p.start - k);
  const char* sbase = (const char*)(t == 0 ? p.k_src_ptrs[layer] : p.v_src_ptrs[layer]);
  const char* sp = sbase + (b * L.src_sb + h * L.src_sh) * p.es + (int64_t)src_row * p.row_bytes + piece * 16;
  o.doff = (b * L.dst_sb + h * L.dst_sh) * p.es + (int64_t)r * p.row_bytes + piece * 16;
  o.lo_v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp));
  o.hi_v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp + half_bytes));
  return o;
}
template <typename T>
__device__ __forceinline__ void ragged_store(const RaggedParams& p, const int layer, const int t, const int tid, const RaggedRow& o) {
  if (!o.live) return;
  const int rloc = tid / p.half_ppr, piece = tid - rloc * p.half_ppr;
  const int half_bytes = p.row_bytes >> 1, r = o.r;
  const int64_t doff = o.doff;
  const u32x4 lo_v = o.lo_v, hi_v = o.hi_v;
  char* dbase = (char*)(t == 0 ? p.k_dst_ptrs[layer] : p.v_dst_ptrs[layer]);
  __builtin_nontemporal_store(lo_v, reinterpret_cast<u32x4*>(dbase + doff));
  __builtin_nontemporal_store(hi_v, reinterpret_cast<u32x4*>(dbase + doff + half_bytes));
  if (t == 0 && p.kr_dst_ptrs) {       // the rotated shadow of the new cache: row r at position r (modify_llama.py:103-104)
    const int pos = min(r, p.table_rows - 1);
    const int64_t toff = (int64_t)pos * half_bytes + piece * 16;
    const u32x4 cs = *reinterpret_cast<const u32x4*>((const char*)p.cos + toff);
    const u32x4 sn = *reinterpret_cast<const u32x4*>((const char*)p.sin + toff);
    constexpr int E = 16 / sizeof(T);
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
    char* rbase = (char*)p.kr_dst_ptrs[layer];
    __builtin_nontemporal_store(olo, reinterpret_cast<u32x4*>(rbase + doff));
    __builtin_nontemporal_store(ohi, reinterpret_cast<u32x4*>(rbase + doff + half_bytes));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kv_compact_ragged_kernel(const RaggedParams p) {
  const int layer = p.l0 + (int)(blockIdx.z >> 1), t = blockIdx.z & 1;
  const LayerPrune L = p.lay[layer];
  const RaggedRow row = ragged_load<T, false>(p, L, layer, t, (int)blockIdx.y, (int)blockIdx.x, (int)threadIdx.x);
  ragged_store<T>(p, layer, t, (int)threadIdx.x, row);
}

__global__ __launch_bounds__(256) void acc_compact_ragged_kernel(const LayerPrune* __restrict__ lay,
                                                                 const float* const* __restrict__ src_ptrs,
                                                                 float* const* __restrict__ dst_ptrs,
                                                                 const int32_t* __restrict__ idx, int64_t idx_sl, int64_t idx_sh,
                                                                 int start, int l0) {
  const int r = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, l = l0 + (int)blockIdx.z;
  const LayerPrune L = lay[l];
  if (r >= (int)L.new_len) return;
  const int k = (int)L.k;
  const int j = r < start ? r : (r < start + k ? idx[l * idx_sl + h * idx_sh + (r - start)] : (int)L.hi + (r - start - k));
  dst_ptrs[l][h * L.acc_dst_sh + r] = src_ptrs[l][h * L.acc_src_sh + j];
}

}  // namespace spatten

using namespace spatten;

// the library-owned side stream of the layer-cascade event (one per device, created on first use, never destroyed: like the
// communicator of comm.hip it lives as long as the process)
struct SideStream {
  static constexpr int kEvents = 16;
  hipStream_t s = nullptr;
  hipEvent_t ev[kEvents] = {};
  hipEvent_t join = nullptr;
};
static std::mutex g_side_mutex;      // creation of the side streams, and the issue of one event's legs (its events are shared)
static SideStream* side_stream() {   // (called with g_side_mutex held)
  static SideStream per_dev[16];
  static bool failed[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || failed[dev]) return nullptr;
  SideStream& S = per_dev[dev];
  if (S.s) return &S;
  if (hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking) != hipSuccess) { S.s = nullptr; failed[dev] = true; return nullptr; }
  bool ok = hipEventCreateWithFlags(&S.join, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < SideStream::kEvents && ok; ++i) ok = hipEventCreateWithFlags(&S.ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { failed[dev] = true; return nullptr; }
  return &S;
}

#ifdef SPATTEN_LC_TRACE
extern "C" int spatten_debug_set_lc_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lc_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

static int prune_layer_cascade_impl(int score_dtype, int kv_dtype, int layers, const void* lay_dev, const void* lay_host,
                                    const void* const* score_ptrs, const int32_t* const* known_ptrs,
                                    int32_t* const* new_ids_ptrs, const void* const* k_src_ptrs,
                                    const void* const* v_src_ptrs, void* const* k_dst_ptrs, void* const* v_dst_ptrs,
                                    void* const* kr_dst_ptrs, const void* cos, const void* sin, int table_rows,
                                    int32_t* idx, int kmax, uint32_t* key_scratch, int64_t key_scratch_sh,
                                    const float* const* acc_src_ptrs, float* const* acc_dst_ptrs, int batch, int heads,
                                    int head_dim, int start, unsigned* sync_words, unsigned generation, void* stream);

extern "C" int spatten_prune_layer_cascade(int score_dtype, int kv_dtype, int layers, const void* lay_dev, const void* lay_host,
                                           const void* const* score_ptrs, const int32_t* const* known_ptrs,
                                           int32_t* const* new_ids_ptrs, const void* const* k_src_ptrs,
                                           const void* const* v_src_ptrs, void* const* k_dst_ptrs, void* const* v_dst_ptrs,
                                           void* const* kr_dst_ptrs, const void* cos, const void* sin, int table_rows,
                                           int32_t* idx, int kmax, uint32_t* key_scratch, int64_t key_scratch_sh,
                                           const float* const* acc_src_ptrs, float* const* acc_dst_ptrs, int batch, int heads,
                                           int head_dim, int start, void* stream) {
  return prune_layer_cascade_impl(score_dtype, kv_dtype, layers, lay_dev, lay_host, score_ptrs, known_ptrs, new_ids_ptrs, k_src_ptrs,
                                  v_src_ptrs, k_dst_ptrs, v_dst_ptrs, kr_dst_ptrs, cos, sin, table_rows, idx, kmax, key_scratch,
                                  key_scratch_sh, acc_src_ptrs, acc_dst_ptrs, batch, heads, head_dim, start, nullptr, 0u, stream);
}

static int prune_layer_cascade_impl(int score_dtype, int kv_dtype, int layers, const void* lay_dev, const void* lay_host,
                                           const void* const* score_ptrs, const int32_t* const* known_ptrs,
                                           int32_t* const* new_ids_ptrs, const void* const* k_src_ptrs,
                                           const void* const* v_src_ptrs, void* const* k_dst_ptrs, void* const* v_dst_ptrs,
                                           void* const* kr_dst_ptrs, const void* cos, const void* sin, int table_rows,
                                           int32_t* idx, int kmax, uint32_t* key_scratch, int64_t key_scratch_sh,
                                    const float* const* acc_src_ptrs, float* const* acc_dst_ptrs, int batch, int heads,
                                    int head_dim, int start, unsigned* sync_words, unsigned generation, void* stream) {
  if (!ok_dtype(score_dtype) || !ok_dtype(kv_dtype) || layers <= 0 || !lay_dev || !lay_host || !score_ptrs || !known_ptrs ||
      !new_ids_ptrs || !k_src_ptrs || !v_src_ptrs || !k_dst_ptrs || !v_dst_ptrs || !idx || !key_scratch || batch <= 0 ||
      heads <= 0 || start < 0 || kmax <= 0)
    return SPATTEN_ERR_INVALID;
  if ((acc_src_ptrs == nullptr) != (acc_dst_ptrs == nullptr)) return SPATTEN_ERR_INVALID;
  if (kr_dst_ptrs && (!cos || !sin)) return SPATTEN_ERR_INVALID;
  const LayerPrune* H_ = (const LayerPrune*)lay_host;
  int64_t max_new = 0, prev_k = -1;
  for (int l = 0; l < layers; ++l) {
    const LayerPrune& L = H_[l];
    if (L.len <= 0 || L.hi > L.len || L.hi < start || L.k <= 0 || L.k > kmax || L.new_len != start + L.k + (L.len - L.hi) ||
        L.n_known < 0 || L.n_known > L.len)
      return SPATTEN_ERR_INVALID;
    if (L.hi - start < L.k) return SPATTEN_ERR_WINDOW;
    if (L.hi - start > key_scratch_sh) return SPATTEN_ERR_INVALID;
    if (kr_dst_ptrs && table_rows < L.new_len) return SPATTEN_ERR_INVALID;
    if (prev_k >= 0 && L.k > prev_k) return SPATTEN_ERR_INVALID;       // the surviving set does not grow through the layers
    prev_k = L.k;
    max_new = L.new_len > max_new ? L.new_len : max_new;
  }
  hipStream_t st = (hipStream_t)stream;
  ChainParams c{};
  c.lay = (const LayerPrune*)lay_dev; c.score_ptrs = score_ptrs; c.known_ptrs = known_ptrs; c.new_ids_ptrs = new_ids_ptrs;
  c.idx = idx; c.idx_sl = (int64_t)heads * kmax; c.idx_sh = kmax; c.keys = key_scratch; c.keys_sh = key_scratch_sh;
  c.layers = layers; c.start = start; c.heads = heads;
  int64_t max_w = 0;
  for (int l = 0; l < layers; ++l) max_w = std::max(max_w, H_[l].hi - start);
  // dynamic LDS: the window's keys (up to 60 KB; longer windows use the global scratch), then the membership map — a byte per
  // token id, up to 96 KB (the 160-KB LDS of a gfx950 CU is this one workgroup's: ~17 KB static + at most 140 KB here)
  c.lds_keys = (int)std::min<int64_t>(max_w, 15360);
  c.map_ids = layers <= 254 ? (int)std::min<int64_t>(96 * 1024, 140 * 1024 - (int64_t)c.lds_keys * 4) & ~3 : 0;
  const size_t lds = (size_t)c.lds_keys * sizeof(uint32_t) + (size_t)c.map_ids;
  {
    static bool attr_set[64][4] = {};     // per DEVICE and score dtype (the attribute is a per-device property: ADVICE r04)
    const int di = score_dtype == SPATTEN_F32 ? 0 : (score_dtype == SPATTEN_BF16 ? 1 : 2);
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;
    if (!attr_set[dev_][di]) {
      hipError_t e = hipSuccess;
      SPATTEN_BY_DTYPE(score_dtype, e = hipFuncSetAttribute((const void*)layer_cascade_select_kernel<T>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
      if (e != hipSuccess) { (void)hipGetLastError(); return SPATTEN_ERR_UNSUPPORTED; }
      attr_set[dev_][di] = true;
    }
  }
  const int es = kv_dtype == SPATTEN_F32 ? 4 : 2;
  RaggedParams r{};
  r.lay = (const LayerPrune*)lay_dev; r.k_src_ptrs = k_src_ptrs; r.v_src_ptrs = v_src_ptrs; r.k_dst_ptrs = k_dst_ptrs;
  r.v_dst_ptrs = v_dst_ptrs; r.kr_dst_ptrs = kr_dst_ptrs; r.cos = cos; r.sin = sin; r.table_rows = table_rows;
  r.idx = idx; r.idx_sl = (int64_t)heads * kmax; r.idx_sh = kmax;
  r.B = batch; r.H = heads; r.start = start; r.row_bytes = head_dim * es; r.es = es;
  if (r.row_bytes % 32 != 0) return SPATTEN_ERR_UNSUPPORTED;
  r.half_ppr = r.row_bytes / 32;
  if (r.half_ppr > 256) return SPATTEN_ERR_UNSUPPORTED;
  r.rows_per_block = 256 / r.half_ppr;
  if ((long long)batch * heads > 65535 || 2 * layers > 65535) return SPATTEN_ERR_UNSUPPORTED;
  (void)sync_words; (void)generation;
  // The chain is ONE workgroup per head for the whole event and the gather cannot start before its layer's positions exist, but
  // layer l's gather does not have to wait for layer l + 1's selection: the event is cut into kLegs legs of consecutive layers,
  // the chain walks them as separate launches on the caller's stream and each leg's gather runs on a library-owned side stream
  // behind an event — a full-speed, free-running grid UNDER the next leg's (latency-bound, 32-workgroup) chain; the caller's
  // stream joins the side stream at the end (a fork / join that a stream capture follows as well).  r04: the same overlap
  // inside one launch (persistent gather workers polling per-layer words) lost; as launches it wins because the gather
  // keeps its own grid.  SPATTEN_LC_LEGS=1 is the serial form.
  static int env_legs = -1;
  if (env_legs < 0) { const char* e = getenv("SPATTEN_LC_LEGS"); env_legs = e ? std::max(1, atoi(e)) : 4; }
  const int legs = std::min(std::max(env_legs, ceil_div(layers, kChainMaxLayers)), layers);
  // two host threads issuing events on one device would re-record each other's fork events: one event's legs at a time
  std::unique_lock<std::mutex> side_lock(g_side_mutex, std::defer_lock);
  if (legs > 1) side_lock.lock();
  SideStream* side = legs > 1 ? side_stream() : nullptr;
  const int n_legs = side ? legs : ceil_div(layers, kChainMaxLayers);
  auto launch_gather = [&](int l0, int l1, hipStream_t s) -> int {
    r.l0 = l0;
    int64_t leg_new = 0;
    for (int l = l0; l < l1; ++l) leg_new = std::max(leg_new, H_[l].new_len);
    const dim3 grid((unsigned)ceil_div((int)leg_new, r.rows_per_block), (unsigned)(batch * heads), (unsigned)(2 * (l1 - l0)));
    SPATTEN_BY_DTYPE(kv_dtype, hipLaunchKernelGGL((kv_compact_ragged_kernel<T>), grid, dim3(256), 0, s, r));
    if (hipGetLastError() != hipSuccess) return SPATTEN_ERR_LAUNCH;
    if (acc_src_ptrs) {
      hipLaunchKernelGGL(acc_compact_ragged_kernel, dim3((unsigned)ceil_div((int)leg_new, 256), (unsigned)heads, (unsigned)(l1 - l0)),
                         dim3(256), 0, s, (const LayerPrune*)lay_dev, acc_src_ptrs, acc_dst_ptrs, idx, (int64_t)heads * kmax,
                         (int64_t)kmax, start, l0);
      if (hipGetLastError() != hipSuccess) return SPATTEN_ERR_LAUNCH;
    }
    return SPATTEN_OK;
  };
  // the blocked chain (keys in registers, 4 barriers per layer) where every window fits 4 positions per thread
  static int env_blk = -1;
  if (env_blk < 0) { const char* e = getenv("SPATTEN_LC_BLOCKED"); env_blk = e ? atoi(e) : 1; }
  const bool blocked = env_blk != 0 && max_w <= kBlk * kChainThreads && c.map_ids > 0;
  if (blocked) {
    static bool attr_blk[4] = {};
    const int di = score_dtype == SPATTEN_F32 ? 0 : (score_dtype == SPATTEN_BF16 ? 1 : 2);
    if (!attr_blk[di]) {
      hipError_t e = hipSuccess;
      SPATTEN_BY_DTYPE(score_dtype, e = hipFuncSetAttribute((const void*)layer_cascade_select_blocked_kernel<T>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));   // = the map
      if (e != hipSuccess) return SPATTEN_ERR_LAUNCH;
      attr_blk[di] = true;
    }
  }
  bool forked = false;
  // whatever happens after the fork, the caller's stream joins the side stream again (an unjoined fork would also invalidate a
  // stream capture)
  auto join = [&]() -> bool {
    if (!forked) return true;
    forked = false;
    return hipEventRecord(side->join, side->s) == hipSuccess && hipStreamWaitEvent(st, side->join, 0) == hipSuccess;
  };
  for (int g = 0; g < n_legs; ++g) {
    const int l0 = (int)((int64_t)layers * g / n_legs), l1 = (int)((int64_t)layers * (g + 1) / n_legs);
    c.l_begin = l0; c.l_end = l1;
    if (blocked) {
      SPATTEN_BY_DTYPE(score_dtype, hipLaunchKernelGGL((layer_cascade_select_blocked_kernel<T>), dim3((unsigned)heads),
                                                       dim3(kChainThreads), (size_t)c.map_ids, st, c));
    } else {
      SPATTEN_BY_DTYPE(score_dtype, hipLaunchKernelGGL((layer_cascade_select_kernel<T>), dim3((unsigned)heads), dim3(kChainThreads),
                                                       lds, st, c));
    }
    if (hipGetLastError() != hipSuccess) { join(); return SPATTEN_ERR_LAUNCH; }
    int rc;
    if (g + 1 < n_legs) {       // this leg's gather goes to the side stream, behind the leg's chain
      if (hipEventRecord(side->ev[g % SideStream::kEvents], st) != hipSuccess ||
          hipStreamWaitEvent(side->s, side->ev[g % SideStream::kEvents], 0) != hipSuccess) { join(); return SPATTEN_ERR_LAUNCH; }
      forked = true;
      rc = launch_gather(l0, l1, side->s);
    } else {
      rc = launch_gather(l0, l1, st);
    }
    if (rc != SPATTEN_OK) { join(); return rc; }
  }
  if (!join()) return SPATTEN_ERR_LAUNCH;
  return SPATTEN_OK;
}
