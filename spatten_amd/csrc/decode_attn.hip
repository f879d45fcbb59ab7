// decode_attn.hip — fused single-token (or few-token) attention over a per-head-pruned KV cache.
//
// Replaces modify_llama.py:86-147 at small q_len (reference = ~25 eager torch ops per layer):
//   RoPE(Q @ pos_q) · RoPE(K_j @ j)^T / sqrt(d)  -> stash (pre-mask)  -> +mask -> fp32 softmax -> P·V
// with the new token's K/V row appended in place (replaces torch.cat, :95-98).
//
// Rotated shadow.  The reference re-rotates the WHOLE un-rotated K cache on every step because key
// positions are cache-relative (:100-104).  Between two prune events the position of a cached key (its
// slot) never changes, so its rotated row is constant: the cache keeps, next to the un-rotated K the
// reference's API hands around, a rotated shadow Kr (same layout, model dtype, rounded exactly like the
// reference's three RoPE ops).  Decode streams Kr and V once and touches K only to append the new row;
// the shadow is rebuilt (rope kernel / fused into the compaction) after a prune, when slots move.
// This trades HBM capacity (3 instead of 2 planes; 288 GB per MI355X) for ~4x less VALU work and no
// rotary-table traffic in the per-token kernel.
//
// Shape of the problem (C2: B=1, H=32, d=128, n=2048 kept rows, bf16): 32 MiB of Kr+V per layer, 67 MFLOP
// -> HBM-bound (1 FLOP/B).  No MFMA: a 1xd by dxn product has no reuse to feed a matrix core with.
//
// Mapping (wave64, 256-thread workgroups):
//   grid = (B*H*S, n_q) workgroups; workgroup (b,h,s) owns `chunk` consecutive keys of head h (split-N so
//   that 32 heads still fill 256 CUs).  A row of d elements is covered by LPR = d/16 lanes; lane c holds
//   elements [8c,8c+8) and [d/2+8c, d/2+8c+8) — the two halves RoPE pairs up — so rotating the query and
//   the appended key needs no cross-lane traffic, and every global access is a 16-byte load of a fully
//   used 128-byte line (rows are contiguous, pitch d).
//   A tile = UNR row-groups; all loads of a tile are issued before the first use.  Softmax is online
//   across tiles; per-split partials (o, m, l) go to a small fp32 workspace and the LAST split to arrive
//   for a (b,h) merges them (ticket counter; write-through stores + agent-scope loads, one round trip).
//
// Rounding: in the 16-bit dtypes the reference rounds after every torch op; everything that feeds the
// stash is reproduced (rotations, matmul result -> dtype, /sqrt(d) -> dtype) so stash values — the input
// of the top-k — match the reference except where fp32 accumulation ORDER moves a value across a
// rounding boundary.  P is NOT rounded to the model dtype before P·V (it would need the global softmax
// denominator before the first V row); tolerance stated in tests/util.py.
#include <stdlib.h>

#include "common.h"

#ifdef SPATTEN_TRACE   // developer instrumentation: per-workgroup phase timestamps (tools/mb/decode_trace.cpp)
__device__ unsigned long long* g_spatten_trace = nullptr;
#define SPATTEN_TSTAMP(slot)                                                                         \
  do {                                                                                               \
    if (g_spatten_trace && threadIdx.x == 0)                                                         \
      g_spatten_trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = \
          __builtin_readcyclecounter();                                                              \
  } while (0)
#else
#define SPATTEN_TSTAMP(slot)
#endif

namespace spatten {

template <typename T>
struct DecodeParams {
  const T* q; int64_t q_sb, q_sh, q_sq;
  T* kc; T* krc; T* vc; int64_t kv_sb, kv_sh;
  const T* k_new; const T* v_new; int64_t new_sb, new_sh;
  const T* cos; const T* sin; int table_rows;
  const int64_t* pos_ids; int64_t pos_sb;
  const T* mask; int64_t mask_sb, mask_sq;
  T* out; int64_t out_sb, out_sq;
  T* scores; int64_t sc_sb, sc_sh, sc_sq;
  float* lse;
  const int32_t* head_ids;   // optional: blockIdx.y -> query head (head pruning: only the kept heads are launched)
  const float* scores_in; int64_t si_sb, si_sh;   // optional: final fp32 logits given: no K traffic
  // progressive-quant key planes (KSRC != 0, pq.hip): 4-bit MSB / LSB planes [B,Hkv,cap,D/2] + per-row scale
  const uint8_t* pq_msb; const uint8_t* pq_lsb; const float* pq_scale; int64_t pl_sb, pl_sh, ps_sb, ps_sh;
  float pq_thr; int32_t* pq_need;   // [B*H]: written by the MSB pass (max prob < thr), read by the refetch pass
  unsigned long long* ws_part;   // [B*H*n_q, S, D+2] {value, tag} granules
  unsigned* ws_cnt;     // [B*H*n_q][2]: {arrival counter, launch generation}
  int B, H, Hkv, N, pos_q, S, chunk, n_q, causal;
  float sqrt_d;
};

constexpr int kDecodeThreads = 256;

// one 8-byte {value, tag} granule of a published partial (tag != 0 <=> the value has landed)
__device__ inline void store_granule(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: the fused decode step.  MODE 1 (scores only): stash + (max, sum), no V traffic, no output — first pass of
// local V pruning.  MODE 2 (scores in): the final fp32 logits are given, no K traffic — last pass of the
// progressive-quantisation path.  Compile-time so the hot instantiation carries no extra branches.
// LEAN: the plain decode step (one query row, MHA, no mask / position tensor / head list) — the common case gets an
// instantiation that reads fewer kernel arguments (one scalar-load batch instead of three dependent ones: ~1 us of
// launch-to-first-load latency on a 14 us kernel) and carries no integer divisions.
// KSRC: where keys come from.  0 = the rotated shadow (model dtype).  1 = progressive-quant MSB plane only (pass 1:
// 4 bits / element + a per-row scale; softmax + P·V run speculatively on these logits and the merge step records
// need_lsb = max prob < threshold, RequantDecision.scala:44-72).  2 = MSB | LSB planes (the refetch pass: only the
// heads pass 1 flagged do any work; they recompute the row ONCE at 8 bits, SpAttenController.scala:402).
// (Measured and dropped: a software pipeline over the tiles — loads of tile t+1 issued before the arithmetic of tile t
//  from a second register set — changes nothing at N = 4096 / 8192: the streaming phase already runs at the HBM rate,
//  the rest of the kernel time is launch + first-byte latency + the merge tail.)
// NT: K/V rows are fetched with the non-temporal cache policy (each row is used once per launch: -0.5 us of 13.7 at
// C2); off when several query rows (the rows leg of prefill) re-read the same K/V through L2.
template <typename T, int D, int UNR, int MODE = 0, bool LEAN = false, int KSRC = 0, bool NT = LEAN>
__device__ __forceinline__ void decode_body(const DecodeParams<T>& p) {
  constexpr bool SCORES_ONLY = (MODE == 1);
  constexpr bool SCORES_IN = (MODE == 2);
  constexpr bool PQ = (KSRC != 0);
  constexpr int LPR = D / 16;                    // lanes per row
  constexpr int RPI = kDecodeThreads / LPR;      // rows per iteration of the workgroup
  constexpr int TILE = RPI * UNR;
  constexpr int HALF = D / 2;
  constexpr int G = (kDecodeThreads / D) > 0 ? (kDecodeThreads / D) : 1;   // merge thread groups
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  using D8 = Dot8<T>;

  __shared__ float s_o[4][D + 2];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x;
  const int c = tid % LPR;
  const int r = tid / LPR;
  const int wave = tid / kWave;
  const int lane = tid % kWave;
  SPATTEN_TSTAMP(0);

  // grid = (S, H, B * n_q): no integer divisions on the way to the first load
  const int split = blockIdx.x;
  const int h = (!LEAN && p.head_ids) ? p.head_ids[blockIdx.y] : (int)blockIdx.y;
  const int b = (LEAN || p.n_q == 1) ? (int)blockIdx.z : (int)blockIdx.z / p.n_q;
  const int qi = (LEAN || p.n_q == 1) ? 0 : (int)blockIdx.z - b * p.n_q;
  const int hkv = (LEAN || p.Hkv == p.H) ? h : h / (p.H / p.Hkv);
  const int unit = LEAN ? (b * p.H + h) : (b * p.H + h) * p.n_q + qi;   // one softmax row
  if (KSRC == 2 && p.pq_need[unit] == 0) return;   // confident head: the MSB pass already produced its output

  // keys this query may attend to (HF causal: j <= P + i with P = N - n_q); the stash covers all N
  const int n_vis = (!LEAN && p.causal) ? min(p.N, p.N - p.n_q + qi + 1) : p.N;
  const int lo = split * p.chunk;
  const int hi = min(lo + p.chunk, (p.scores != nullptr) ? p.N : n_vis);

  T* krbase = p.krc + b * p.kv_sb + hkv * p.kv_sh;
  T* vbase = p.vc + b * p.kv_sb + hkv * p.kv_sh;
  const bool owns_new = (p.k_new != nullptr) && lo < p.N && hi == p.N;   // this workgroup appends row N-1

  // ---- loads of the first tile go out before anything else --------------------------------------
  struct Tile {
    raw_t k_lo[UNR], k_hi[UNR], v_lo[UNR], v_hi[UNR];
    uint32_t pm_lo[UNR], pm_hi[UNR], pl_lo[UNR], pl_hi[UNR];   // PQ: 8 nibbles each (elements [8c,8c+8) / [d/2+8c, ..))
    float pscale[UNR];
  };
  Tile tile_a;
  const uint8_t* pq_m = PQ ? p.pq_msb + b * p.pl_sb + hkv * p.pl_sh : nullptr;
  const uint8_t* pq_l = KSRC == 2 ? p.pq_lsb + b * p.pl_sb + hkv * p.pl_sh : nullptr;
  const float* pq_s = PQ ? p.pq_scale + b * p.ps_sb + hkv * p.ps_sh : nullptr;
  auto issue_tile = [&](Tile& tl, int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      int j = t0 + u * RPI + r;
      j = j < hi ? j : hi - 1;
      if (PQ) {
        const int64_t po = (int64_t)j * HALF + 4 * c;   // a plane row is D/2 bytes
        tl.pm_lo[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pq_m + po));
        tl.pm_hi[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pq_m + po + HALF / 2));
        if (KSRC == 2) {
          tl.pl_lo[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pq_l + po));
          tl.pl_hi[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pq_l + po + HALF / 2));
        }
        tl.pscale[u] = pq_s[j];
      }
      const T* kp = krbase + (int64_t)j * D;
      const T* vp = vbase + (int64_t)j * D;
      if (owns_new && j == p.N - 1) {   // the token being appended: source = k_new / v_new (un-rotated)
        kp = p.k_new + b * p.new_sb + hkv * p.new_sh;
        vp = p.v_new + b * p.new_sb + hkv * p.new_sh;
      }
      if (!SCORES_IN && !PQ) {
        tl.k_lo[u] = NT ? V8::ldg_stream(kp + 8 * c) : V8::ldg(kp + 8 * c);
        tl.k_hi[u] = NT ? V8::ldg_stream(kp + HALF + 8 * c) : V8::ldg(kp + HALF + 8 * c);
      }
      if (!SCORES_ONLY || owns_new) {
        tl.v_lo[u] = NT ? V8::ldg_stream(vp + 8 * c) : V8::ldg(vp + 8 * c);
        tl.v_hi[u] = NT ? V8::ldg_stream(vp + HALF + 8 * c) : V8::ldg(vp + HALF + 8 * c);
      }
    }
  };
  if (lo < hi) issue_tile(tile_a, lo);
  SPATTEN_TSTAMP(5);
  // this unit's launch generation (tags of the published partials, see below): written by the previous launch's merger,
  // so it comes from memory — fetched with a VECTOR load queued behind the first tile (a scalar load would be waited
  // for with the kernel arguments, 2-3 us before anything else happens)
  const unsigned gen = p.S > 1 ? p.ws_cnt[2 * unit + 1 + opaque_lane(0)] : 0u;

  // ---- un-rotated query + its table row ----------------------------------------------------------
  typename D8::packed q_lo, q_hi;                // rotated query, packed in the model dtype (exact: it IS rounded)
  typename NibbleDot<T>::packed qn_lo, qn_hi;    // PQ: the same rotated query, arranged for the nibble dot product
  raw_t n_raw[2];
  {
    const T* qp = p.q + b * p.q_sb + h * p.q_sh + (LEAN ? 0 : qi * p.q_sq);
    int pq = (!LEAN && p.pos_ids) ? (int)p.pos_ids[b * p.pos_sb + qi] : p.pos_q + qi;
    pq = min(max(pq, 0), p.table_rows - 1);
    const raw_t q0 = V8::ldg(qp + 8 * c);
    const raw_t q1 = V8::ldg(qp + HALF + 8 * c);
    const raw_t q2 = V8::ldg(p.cos + (int64_t)pq * HALF + 8 * c);
    const raw_t q3 = V8::ldg(p.sin + (int64_t)pq * HALF + 8 * c);
    if (owns_new) {   // the appended key is rotated at its slot index N-1 (modify_llama.py:103-104)
      n_raw[0] = V8::ldg(p.cos + (int64_t)(p.N - 1) * HALF + 8 * c);
      n_raw[1] = V8::ldg(p.sin + (int64_t)(p.N - 1) * HALF + 8 * c);
    }
    float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
    V8::unpack(q0, xlo);
    V8::unpack(q1, xhi);
    V8::unpack(q2, cc);
    V8::unpack(q3, ss);
    rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
    q_lo = D8::pack(ylo);
    q_hi = D8::pack(yhi);
    if (PQ) { qn_lo = NibbleDot<T>::prep(ylo); qn_hi = NibbleDot<T>::prep(yhi); }
  }
  SPATTEN_TSTAMP(6);
  const float rsqrt_d = 1.0f / p.sqrt_d;
  const T* maskp = (!LEAN && p.mask) ? p.mask + b * p.mask_sb + qi * p.mask_sq : nullptr;
  T* stashp = p.scores ? p.scores + b * p.sc_sb + h * p.sc_sh + (LEAN ? 0 : qi * p.sc_sq) : nullptr;
  T* kbase = p.kc ? p.kc + b * p.kv_sb + hkv * p.kv_sh : nullptr;

  // per-THREAD online softmax (the LPR lanes of a row share its score, so they agree): no barrier and
  // no cross-lane maximum inside the loop; the row groups are reconciled once, after the loop.
  float m_run = -INFINITY, l_run = 0.f;
  float olo[8], ohi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = 0.f; ohi[i] = 0.f; }

  auto process_tile = [&](Tile& tl, int t0) {
    float mk[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) mk[u] = maskp ? DT<T>::to_f32(maskp[min(t0 + u * RPI + r, n_vis - 1)]) : 0.f;

    if (owns_new && t0 + TILE >= p.N) {
      // append in place: K un-rotated (modify_llama.py:95-100), its rotation into the shadow, V
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int j = t0 + u * RPI + r;
        if (j == p.N - 1) {
          float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
          V8::unpack(tl.k_lo[u], xlo);
          V8::unpack(tl.k_hi[u], xhi);
          V8::unpack(n_raw[0], cc);
          V8::unpack(n_raw[1], ss);
          rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
          if (kbase) {
            V8::stg(kbase + (int64_t)j * D + 8 * c, tl.k_lo[u]);
            V8::stg(kbase + (int64_t)j * D + HALF + 8 * c, tl.k_hi[u]);
          }
          tl.k_lo[u] = V8::pack(ylo);
          tl.k_hi[u] = V8::pack(yhi);
          V8::stg(krbase + (int64_t)j * D + 8 * c, tl.k_lo[u]);
          V8::stg(krbase + (int64_t)j * D + HALF + 8 * c, tl.k_hi[u]);
          V8::stg(vbase + (int64_t)j * D + 8 * c, tl.v_lo[u]);
          V8::stg(vbase + (int64_t)j * D + HALF + 8 * c, tl.v_hi[u]);
        }
      }
    }

    // ---- scores of the UNR row groups as independent instruction streams (ILP: one wave per SIMD has nothing
    // else to hide a dependent VALU chain behind), then ONE per-thread softmax update, then P·V ------------
    float sc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (SCORES_IN) sc[u] = p.scores_in[b * p.si_sb + h * p.si_sh + min(t0 + u * RPI + r, hi - 1)];
      else if (PQ) {
        // sum_i q_i * q8_i with q8 = 16 * sext(msb nibble) + lsb nibble; the row scale is applied after the reduction
        float a = NibbleDot<T>::dot(qn_lo, tl.pm_lo[u] ^ 0x88888888u, 8.f) + NibbleDot<T>::dot(qn_hi, tl.pm_hi[u] ^ 0x88888888u, 8.f);
        a *= 16.f;
        if (KSRC == 2) a += NibbleDot<T>::dot(qn_lo, tl.pl_lo[u], 0.f) + NibbleDot<T>::dot(qn_hi, tl.pl_hi[u], 0.f);
        sc[u] = a;
      } else sc[u] = D8::dot(q_hi, tl.k_hi[u], D8::dot(q_lo, tl.k_lo[u], 0.f));
    }
    if (PQ) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) sc[u] = group_sum<LPR>(sc[u]) * tl.pscale[u] / p.sqrt_d;   // fp32 logits
    } else if (!SCORES_IN) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) sc[u] = group_sum<LPR>(sc[u]);
      // matmul result -> dtype, then the separate divide -> dtype (modify_llama.py:111-113)
#pragma unroll
      for (int u = 0; u < UNR; ++u) sc[u] = DT<T>::round(div_by_const(DT<T>::round(sc[u]), p.sqrt_d, rsqrt_d));
    }
#ifdef SPATTEN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tracing build only: when has ALL of the tile landed?
    SPATTEN_TSTAMP(7);
#endif
    float m_new = m_run;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = t0 + u * RPI + r;
      const bool valid = j < hi;
      float s = sc[u];
      if (stashp != nullptr && c == 0 && valid) stashp[j] = DT<T>::from_f32(s);       // pre-mask (:116-119)
      if (maskp != nullptr) s = DT<T>::round(s + mk[u]);                               // :132
      s = (valid && j < n_vis) ? s : -INFINITY;
      sc[u] = s;
      m_new = fmaxf(m_new, s);
    }
    if (m_new > m_run) {                           // rare after the first tile
      const float alpha = __expf(m_run - m_new);   // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) { olo[i] *= alpha; ohi[i] *= alpha; }
      m_run = m_new;
    }
    float pj[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      pj[u] = (sc[u] == -INFINITY) ? 0.f : __expf(sc[u] - m_run);
      l_run += pj[u];
    }
    if (!SCORES_ONLY) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float vlo[8], vhi[8];
        V8::unpack(tl.v_lo[u], vlo);
        V8::unpack(tl.v_hi[u], vhi);
#pragma unroll
        for (int i = 0; i < 8; ++i) { olo[i] = fmaf(pj[u], vlo[i], olo[i]); ohi[i] = fmaf(pj[u], vhi[i], ohi[i]); }
      }
    }
  };
  for (int t0 = lo; t0 < hi; t0 += TILE) {
    if (t0 != lo) issue_tile(tile_a, t0);
    process_tile(tile_a, t0);
  }

  SPATTEN_TSTAMP(1);
  // ---- reconcile the row groups: per-wave max and sums (registers only), then ONE LDS hop across the waves ------
  // (measured alternatives: an extra barrier for a workgroup-wide max first — same time; LDS over the 16 DPP rows
  //  instead of the permlane swaps — slower: 16 exps + 48 LDS reads per thread in the last stage)
  {
    const float mw = wave_max(m_run);
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - mw);
    l_run *= alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) { olo[i] *= alpha; ohi[i] *= alpha; }
    m_run = mw;
  }
  // lanes with equal c across the wave's row groups: in-row rotations (DPP), then rows (permlane swaps).
  // (Measured alternative: sending the 16 DPP rows through LDS instead of the swaps is SLOWER — 2.9k vs 1.9k cycles.)
  if (LPR == 4) {
    l_run += dpp_mov<kDppRor8>(l_run);
    l_run += dpp_mov<kDppRor4>(l_run);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      olo[i] += dpp_mov<kDppRor8>(olo[i]); olo[i] += dpp_mov<kDppRor4>(olo[i]);
      ohi[i] += dpp_mov<kDppRor8>(ohi[i]); ohi[i] += dpp_mov<kDppRor4>(ohi[i]);
    }
  } else if (LPR == 8) {
    l_run += dpp_mov<kDppRor8>(l_run);
#pragma unroll
    for (int i = 0; i < 8; ++i) { olo[i] += dpp_mov<kDppRor8>(olo[i]); ohi[i] += dpp_mov<kDppRor8>(ohi[i]); }
  }
  l_run = xor32_sum(xor16_sum(l_run));
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = xor32_sum(xor16_sum(olo[i])); ohi[i] = xor32_sum(xor16_sum(ohi[i])); }
  if (lane < LPR) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_o[wave][8 * lane + i] = olo[i];
      s_o[wave][HALF + 8 * lane + i] = ohi[i];
    }
    if (lane == 0) { s_o[wave][D] = l_run; s_o[wave][D + 1] = m_run; }
  }
  __syncthreads();
  float o_tot = 0.f, l_tot = 0.f;
  {
    const float m0 = s_o[0][D + 1], m1 = s_o[1][D + 1], m2 = s_o[2][D + 1], m3 = s_o[3][D + 1];
    const float m_wg = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float mu = (m_wg == -INFINITY) ? 0.f : m_wg;
    const float w0 = __expf(m0 - mu), w1 = __expf(m1 - mu), w2 = __expf(m2 - mu), w3 = __expf(m3 - mu);   // exp(-inf) = 0
    if (tid < D) o_tot = (s_o[0][tid] * w0 + s_o[1][tid] * w1) + (s_o[2][tid] * w2 + s_o[3][tid] * w3);
    l_tot = (s_o[0][D] * w0 + s_o[1][D] * w1) + (s_o[2][D] * w2 + s_o[3][D] * w3);
    m_run = m_wg;
  }
  SPATTEN_TSTAMP(2);
  T* outp = p.out + b * p.out_sb + (LEAN ? 0 : qi * p.out_sq) + h * D;
  if (p.S == 1) {
    if (!SCORES_ONLY && tid < D) outp[tid] = DT<T>::from_f32(o_tot / l_tot);
    if (p.lse != nullptr && tid == 0) { p.lse[unit * 2] = m_run; p.lse[unit * 2 + 1] = l_tot; }
    if (KSRC == 1 && tid == 0) p.pq_need[unit] = (1.0f / l_tot) < p.pq_thr ? 1 : 0;   // max prob = exp(0) / sum
    return;
  }

  // ---- publish the partial; the last split to arrive merges ------------------------------------
  // Every value goes out as ONE 8-byte write-through {value, tag} granule (agent-scope relaxed atomic
  // store = sc1): the data is its own flag, so nobody waits for store acknowledgements.  The ticket
  // tells the last arriver that every other split has ISSUED its granules; it then reads them with
  // agent-scope loads (placement independent across the 8 XCD L2s) and re-reads the rare granule whose
  // tag has not landed yet.  The tag is the unit's launch GENERATION + 1 (a word next to the counter, advanced by
  // the merger): granules of earlier launches never match, so nothing has to be cleared for the next launch
  // (clearing S x (D+2) granules cost the merger 0.3 us of a 13 us kernel).
  unsigned long long* ws = p.ws_part + ((int64_t)unit * p.S) * (D + 2);
  unsigned long long* part = ws + (int64_t)split * (D + 2);
  const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;
  // the ticket is drawn by the LAST wave, which has no stores in flight: on CDNA4 vmcnt also counts stores, so a
  // wave that just issued granules would wait for their write-through acknowledgements before it sees its ticket
  if (tid == kDecodeThreads - 1)
    s_ticket = __hip_atomic_fetch_add(p.ws_cnt + 2 * unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < D && tid < kDecodeThreads - kWave) store_granule(part + tid, o_tot, tag);
  if (D > kDecodeThreads - kWave && tid >= kDecodeThreads - kWave && tid < D) store_granule(part + tid, o_tot, tag);   // D = 256 only
  if (tid == (D < kDecodeThreads - kWave ? D : 0)) { store_granule(part + D, m_run, tag); store_granule(part + D + 1, l_tot, tag); }
  __syncthreads();
  SPATTEN_TSTAMP(3);
  if (s_ticket != (unsigned)(p.S - 1)) return;

  // merge in ONE round trip: thread (g, e) takes splits s = g, g+G, ...; every load below — its partial-o
  // elements and the (m, l) of the same splits — is independent.  Each group folds its splits relative to
  // its own running max; the groups are then folded through LDS.
  constexpr int KB = 8;                          // splits per thread per round trip
  const int e = tid % D, g = tid / D;
  float mg = -INFINITY, lg = 0.f, og = 0.f;
  if (g < G) {
    for (int s0 = g; s0 < p.S; s0 += KB * G) {
      unsigned long long ga[KB], gm[KB], gl[KB];
      int spins = 0;
      bool landed;
      do {   // every load is issued before any tag is looked at: ONE round trip (a short-circuiting `&&` chain makes
             // the compiler wait for each split's granules before it loads the next split's)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int sc_ = (s0 + k * G) < p.S ? (s0 + k * G) : g;
          const unsigned long long* q = ws + (int64_t)sc_ * (D + 2);
          ga[k] = __hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gm[k] = __hip_atomic_load(q + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gl[k] = __hip_atomic_load(q + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned diff = 0u;                      // bitwise, not &&: all loads stay in one round trip
#pragma unroll
        for (int k = 0; k < KB; ++k)
          diff |= ((unsigned)(ga[k] >> 32) ^ tag) | ((unsigned)(gm[k] >> 32) ^ tag) | ((unsigned)(gl[k] >> 32) ^ tag);
        landed = diff == 0u;
      } while (!landed && ++spins < (1 << 20));   // bounded: a granule that was issued always lands
      float a[KB], ms[KB], ls[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const bool live = (s0 + k * G) < p.S;
        a[k] = live ? __uint_as_float((unsigned)ga[k]) : 0.f;
        ms[k] = live ? __uint_as_float((unsigned)gm[k]) : -INFINITY;
        ls[k] = live ? __uint_as_float((unsigned)gl[k]) : 0.f;
      }
      float mn = mg;
#pragma unroll
      for (int k = 0; k < KB; ++k) mn = fmaxf(mn, ms[k]);
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu);
      og *= w0; lg *= w0;
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const float w = __expf(ms[k] - mu);
        og = fmaf(a[k], w, og);
        lg = fmaf(ls[k], w, lg);
      }
      mg = mn;
    }
  }
  if (G > 1) {                                   // fold the thread groups through LDS
    if (g < G) { s_o[g][e] = og; if (e == 0) { s_o[g][D] = mg; s_o[g][D + 1] = lg; } }
    __syncthreads();
    if (g == 0) {
      float mn = mg;
#pragma unroll
      for (int gg = 1; gg < G; ++gg) mn = fmaxf(mn, s_o[gg][D]);
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu);
      og *= w0; lg *= w0;
#pragma unroll
      for (int gg = 1; gg < G; ++gg) {
        const float w = __expf(s_o[gg][D] - mu);
        og = fmaf(s_o[gg][e], w, og);
        lg = fmaf(s_o[gg][D + 1], w, lg);
      }
      mg = mn;
    }
  }
  if (!SCORES_ONLY && g == 0) outp[e] = DT<T>::from_f32(og / lg);
  if (tid == 0) {
    if (p.lse != nullptr) { p.lse[unit * 2] = mg; p.lse[unit * 2 + 1] = lg; }
    if (KSRC == 1) p.pq_need[unit] = (1.0f / lg) < p.pq_thr ? 1 : 0;
    p.ws_cnt[2 * unit + 1] = gen + 1u;                                                         // next launch: new tag
    __hip_atomic_store(p.ws_cnt + 2 * unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm the counter
  }
  SPATTEN_TSTAMP(4);
}

template <typename T, int D, int UNR, int MODE = 0, bool LEAN = false, int KSRC = 0, bool NT = LEAN>
__global__ __launch_bounds__(kDecodeThreads) void decode_attn_kernel(const DecodeParams<T> p) {
  decode_body<T, D, UNR, MODE, LEAN, KSRC, NT>(p);
}

// The plain decode step with its launch-critical arguments FIRST and 32-bit strides: built with
// -amdgpu-kernarg-preload-count=16 (Makefile) the first 16 kernel-argument dwords arrive in SGPRs with the wave, so
// the first K/V tile loads are issued without waiting for a scalar load of the argument block (the other arguments
// are fetched while those loads are in flight).
template <typename T, int D, int UNR>
__global__ __launch_bounds__(kDecodeThreads) void decode_lean_kernel(T* krc, T* vc, const T* k_new, const T* v_new,
                                                                     int kv_sb, int kv_sh, int new_sb, int new_sh,
                                                                     int N, int chunk, int H, const DecodeParams<T> rest) {
  DecodeParams<T> p = rest;
  p.krc = krc; p.vc = vc; p.k_new = k_new; p.v_new = v_new;
  p.kv_sb = kv_sb; p.kv_sh = kv_sh; p.new_sb = new_sb; p.new_sh = new_sh;
  p.N = N; p.chunk = chunk; p.H = H;
  decode_body<T, D, UNR, 0, true, 0, true>(p);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// row-groups per tile (UNR): tuning knob, overridable with SPATTEN_DECODE_UNR (1|2|4) for experiments
static int decode_unr_for(int dtype) {
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("SPATTEN_DECODE_UNR");
    env = e ? atoi(e) : 0;
  }
  int u = env > 0 ? env : (dtype == SPATTEN_F32 ? 2 : 4);
  if (dtype == SPATTEN_F32 && u > 2) u = 2;
  return (u == 1 || u == 2 || u == 4 || u == 8) ? u : 4;
}
static inline int decode_tile_rows(int d, int dtype) { return (kDecodeThreads / (d / 16)) * decode_unr_for(dtype); }

static int auto_splits(int units, int d, int kv_len, int dtype) {
  const int tile = decode_tile_rows(d, dtype);
  const int max_by_len = ceil_div(kv_len, tile);
  // one workgroup per CU (256 CUs): measured best at Llama-2-7B decode sizes — more splits shorten each
  // workgroup's stream but lengthen the merge (a memory round trip per batch of partials)
  int s = 256 / (units > 0 ? units : 1);
  static int env_s = -1;
  if (env_s < 0) { const char* e = getenv("SPATTEN_DECODE_SPLITS"); env_s = e ? atoi(e) : 0; }
  if (env_s > 0) s = env_s;
  if (s > max_by_len) s = max_by_len;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}

template <typename T, int D>
static int launch_decode(const DecodeParams<T>& p, int n_active, bool scores_only, hipStream_t stream) {
  const dim3 grid((unsigned)p.S, (unsigned)n_active, (unsigned)(p.B * p.n_q));
  if (scores_only) {
    if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((decode_attn_kernel<T, D, 2, 1>), grid, dim3(kDecodeThreads), 0, stream, p);
    else hipLaunchKernelGGL((decode_attn_kernel<T, D, 4, 1>), grid, dim3(kDecodeThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
  }
  if (p.pq_msb != nullptr) {   // progressive-quant keys: pass 1 on the MSB plane, then the refetch pass (same grid)
    if constexpr (D == 256) return SPATTEN_ERR_UNSUPPORTED;
    else {
      constexpr int U = sizeof(T) == 4 ? 2 : 4;
      hipLaunchKernelGGL((decode_attn_kernel<T, D, U, 0, false, 1, true>), grid, dim3(kDecodeThreads), 0, stream, p);
      hipLaunchKernelGGL((decode_attn_kernel<T, D, U, 0, false, 2, true>), grid, dim3(kDecodeThreads), 0, stream, p);
      return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
    }
  }
  if (p.scores_in != nullptr) {
    if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((decode_attn_kernel<T, D, 2, 2>), grid, dim3(kDecodeThreads), 0, stream, p);
    else hipLaunchKernelGGL((decode_attn_kernel<T, D, 4, 2>), grid, dim3(kDecodeThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
  }
  const bool lean = p.n_q == 1 && p.Hkv == p.H && !p.mask && !p.pos_ids && !p.head_ids && !p.causal;
  if (lean && decode_unr_for(DT<T>::kId) == 4) {
    constexpr int U = sizeof(T) == 4 ? 2 : 4;
    const int64_t lim = 0x7FFFFFFF;
    if (p.kv_sb <= lim && p.kv_sh <= lim && p.new_sb <= lim && p.new_sh <= lim)
      hipLaunchKernelGGL((decode_lean_kernel<T, D, U>), grid, dim3(kDecodeThreads), 0, stream, p.krc, p.vc, p.k_new, p.v_new,
                         (int)p.kv_sb, (int)p.kv_sh, (int)p.new_sb, (int)p.new_sh, p.N, p.chunk, p.H, p);
    else
      hipLaunchKernelGGL((decode_attn_kernel<T, D, U, 0, true>), grid, dim3(kDecodeThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
  }
  switch (decode_unr_for(DT<T>::kId)) {
    case 1: hipLaunchKernelGGL((decode_attn_kernel<T, D, 1>), grid, dim3(kDecodeThreads), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((decode_attn_kernel<T, D, 2>), grid, dim3(kDecodeThreads), 0, stream, p); break;
    case 8:
      if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((decode_attn_kernel<T, D, 2>), grid, dim3(kDecodeThreads), 0, stream, p);
      else hipLaunchKernelGGL((decode_attn_kernel<T, D, 8>), grid, dim3(kDecodeThreads), 0, stream, p);
      break;
    default: {
      constexpr int U = sizeof(T) == 4 ? 2 : 4;
      if (p.n_q == 1) hipLaunchKernelGGL((decode_attn_kernel<T, D, U, 0, false, 0, true>), grid, dim3(kDecodeThreads), 0, stream, p);
      else hipLaunchKernelGGL((decode_attn_kernel<T, D, U>), grid, dim3(kDecodeThreads), 0, stream, p);
    }
  }
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

template <typename T>
static int dispatch_decode(DecodeParams<T>& p, int d, int n_active, bool scores_only, hipStream_t stream) {
  switch (d) {
    case 64: return launch_decode<T, 64>(p, n_active, scores_only, stream);
    case 128: return launch_decode<T, 128>(p, n_active, scores_only, stream);
    case 256: return launch_decode<T, 256>(p, n_active, scores_only, stream);
    default: return SPATTEN_ERR_UNSUPPORTED;
  }
}

static size_t decode_cnt_bytes(size_t units) { return (units * 2 * sizeof(unsigned) + 255) / 256 * 256; }

// shared by spatten_attn_decode and the small-q / fp32 leg of spatten_attn_prefill
int decode_rows(int dtype, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq, void* k_cache, void* kr_cache,
                void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* k_new, const void* v_new, int64_t new_sb,
                int64_t new_sh, const void* cos, const void* sin, int table_rows, const int64_t* position_ids,
                int64_t pos_sb, const void* mask, int64_t mask_sb, int64_t mask_sq, void* out, int64_t out_sb,
                int64_t out_sq, void* scores, int64_t sc_sb, int64_t sc_sh, int64_t sc_sq, float* lse, void* workspace,
                size_t workspace_units, int batch, int heads, int kv_heads, int head_dim, int kv_len, int pos_q,
                int n_q, int causal, int n_splits, hipStream_t stream, const int32_t* head_ids, int n_active,
                int flags, const float* scores_in, int64_t si_sb, int64_t si_sh, const PQKeys* pq) {
  const bool scores_only = (flags & SPATTEN_DECODE_SCORES_ONLY) != 0;
  if (!q || (!kr_cache && !scores_in && !pq) || !cos || !sin || (!scores_only && (!out || !v_cache))) return SPATTEN_ERR_INVALID;
  if (scores_only && (!scores || !lse || k_new)) return SPATTEN_ERR_INVALID;
  if (!head_ids) n_active = heads;
  if (n_active <= 0 || n_active > heads) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || kv_len <= 0 || pos_q < 0 || n_q <= 0)
    return SPATTEN_ERR_INVALID;
  if ((k_new == nullptr) != (v_new == nullptr)) return SPATTEN_ERR_INVALID;
  if (k_new && n_q != 1) return SPATTEN_ERR_INVALID;
  if (pq && (!pq->msb || !pq->lsb || !pq->scale || !pq->need || k_new || n_q != 1 || scores_in || scores_only))
    return SPATTEN_ERR_INVALID;
  if (table_rows < kv_len || (!position_ids && pos_q + n_q > table_rows)) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128 && head_dim != 256) return SPATTEN_ERR_UNSUPPORTED;
  if (dtype != SPATTEN_F32 && dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return SPATTEN_ERR_INVALID;
  const int units = batch * heads * n_q;          // workspace is indexed by the FULL head id
  const int tile = decode_tile_rows(head_dim, dtype);
  int S = n_splits > 0 ? n_splits : auto_splits(batch * n_active * n_q, head_dim, kv_len, dtype);
  if (S > ceil_div(kv_len, tile)) S = ceil_div(kv_len, tile);
  if (S > 64) S = 64;
  // chunk = rows per split, a multiple of the tile so every split starts tile-aligned
  const int chunk = ceil_div(ceil_div(kv_len, S), tile) * tile;
  S = ceil_div(kv_len, chunk);
  if (S > 1 && (!workspace || (size_t)units > workspace_units)) return SPATTEN_ERR_INVALID;
  const size_t cnt_bytes = decode_cnt_bytes(workspace_units);

#define SPATTEN_FILL(T)                                                                                  \
  DecodeParams<T> p;                                                                                     \
  p.q = (const T*)q; p.q_sb = q_sb; p.q_sh = q_sh; p.q_sq = q_sq;                                        \
  p.kc = (T*)k_cache; p.krc = (T*)kr_cache; p.vc = (T*)v_cache; p.kv_sb = kv_sb; p.kv_sh = kv_sh;        \
  p.k_new = (const T*)k_new; p.v_new = (const T*)v_new; p.new_sb = new_sb; p.new_sh = new_sh;            \
  p.cos = (const T*)cos; p.sin = (const T*)sin; p.table_rows = table_rows;                               \
  p.pos_ids = position_ids; p.pos_sb = pos_sb;                                                           \
  p.mask = (const T*)mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;                                     \
  p.out = (T*)out; p.out_sb = out_sb; p.out_sq = out_sq;                                                 \
  p.scores = (T*)scores; p.sc_sb = sc_sb; p.sc_sh = sc_sh; p.sc_sq = sc_sq;                              \
  p.lse = lse; p.head_ids = head_ids;                                                                    \
  p.scores_in = scores_in; p.si_sb = si_sb; p.si_sh = si_sh;                                             \
  p.pq_msb = pq ? pq->msb : nullptr; p.pq_lsb = pq ? pq->lsb : nullptr; p.pq_scale = pq ? pq->scale : nullptr; \
  p.pl_sb = pq ? pq->pl_sb : 0; p.pl_sh = pq ? pq->pl_sh : 0; p.ps_sb = pq ? pq->sc_sb : 0; p.ps_sh = pq ? pq->sc_sh : 0; \
  p.pq_thr = pq ? pq->threshold : 0.f; p.pq_need = pq ? pq->need : nullptr;                              \
  p.ws_cnt = (unsigned*)workspace;                                                                       \
  p.ws_part = workspace ? (unsigned long long*)((char*)workspace + cnt_bytes) : nullptr;                              \
  p.B = batch; p.H = heads; p.Hkv = kv_heads; p.N = kv_len; p.pos_q = pos_q; p.S = S; p.chunk = chunk;   \
  p.n_q = n_q; p.causal = causal;                                                                        \
  p.sqrt_d = sqrtf((float)head_dim);                                                                     \
  return dispatch_decode<T>(p, head_dim, n_active, scores_only, stream);

  switch (dtype) {
    case SPATTEN_F32: { SPATTEN_FILL(float) }
    case SPATTEN_F16: { SPATTEN_FILL(f16_t) }
    default: { SPATTEN_FILL(bf16_t) }
  }
#undef SPATTEN_FILL
}

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_decode_workspace_bytes(int batch, int heads, int head_dim, int max_splits) {
  if (batch <= 0 || heads <= 0 || head_dim <= 0 || max_splits <= 0) return 0;
  const size_t units = (size_t)batch * heads;
  return decode_cnt_bytes(units) + units * max_splits * (head_dim + 2) * sizeof(unsigned long long);
}

extern "C" int spatten_decode_auto_splits(int batch, int heads, int head_dim, int kv_len) {
  if (batch <= 0 || heads <= 0 || kv_len <= 0 || (head_dim != 64 && head_dim != 128 && head_dim != 256)) return 1;
  return auto_splits(batch * heads, head_dim, kv_len, SPATTEN_BF16);
}

extern "C" int spatten_attn_decode_ex(int dtype, const void* q, int64_t q_sb, int64_t q_sh, void* k_cache,
                                      void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* k_new,
                                      const void* v_new, int64_t new_sb, int64_t new_sh, const void* cos,
                                      const void* sin, int table_rows, const int64_t* position_ids,
                                      int64_t pos_sb, const void* mask, int64_t mask_sb, void* out,
                                      int64_t out_sb, void* scores, int64_t sc_sb, int64_t sc_sh, float* lse,
                                      void* workspace, int batch, int heads, int kv_heads, int head_dim,
                                      int kv_len, int pos_q, int n_splits, const int32_t* head_ids,
                                      int n_active_heads, int flags, void* stream) {
  return decode_rows(dtype, q, q_sb, q_sh, 0, k_cache, kr_cache, v_cache, kv_sb, kv_sh, k_new, v_new, new_sb, new_sh,
                     cos, sin, table_rows, position_ids, pos_sb, mask, mask_sb, 0, out, out_sb, 0, scores, sc_sb,
                     sc_sh, 0, lse, workspace, (size_t)batch * heads, batch, heads, kv_heads, head_dim, kv_len, pos_q,
                     1, 0, n_splits, (hipStream_t)stream, head_ids, n_active_heads, flags, nullptr, 0, 0, nullptr);
}

extern "C" int spatten_attn_decode(int dtype, const void* q, int64_t q_sb, int64_t q_sh, void* k_cache,
                                   void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* k_new,
                                   const void* v_new, int64_t new_sb, int64_t new_sh, const void* cos,
                                   const void* sin, int table_rows, const int64_t* position_ids,
                                   int64_t pos_sb, const void* mask, int64_t mask_sb, void* out,
                                   int64_t out_sb, void* scores, int64_t sc_sb, int64_t sc_sh, float* lse,
                                   void* workspace, int batch, int heads, int kv_heads, int head_dim,
                                   int kv_len, int pos_q, int n_splits, void* stream) {
  return spatten_attn_decode_ex(dtype, q, q_sb, q_sh, k_cache, kr_cache, v_cache, kv_sb, kv_sh, k_new, v_new, new_sb,
                                new_sh, cos, sin, table_rows, position_ids, pos_sb, mask, mask_sb, out, out_sb, scores,
                                sc_sb, sc_sh, lse, workspace, batch, heads, kv_heads, head_dim, kv_len, pos_q, n_splits,
                                nullptr, 0, 0, stream);
}

#ifdef SPATTEN_TRACE
extern "C" int spatten_debug_set_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_spatten_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
