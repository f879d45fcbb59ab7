// decode_body.h — the body of the decode-attention kernel family (one softmax row per workgroup column, split-N, fused
// append / stash / merge), shared by decode_attn.hip (the per-layer launches) and decode_chain.hip (the chained launch over
// the layers of a token, round 6).  See decode_attn.hip for the design notes.
#pragma once
#include <stdlib.h>

#include <algorithm>

#include <atomic>

#include "common.h"

#ifdef SPATTEN_TRACE   // developer instrumentation: per-workgroup phase timestamps (tools/mb/decode_trace.cpp)
__device__ unsigned long long* g_spatten_trace = nullptr;
#ifndef SPATTEN_TRACE_SLOTS
#define SPATTEN_TRACE_SLOTS 8
#endif
#define SPATTEN_TSTAMP_T(slot, thread)                                                               \
  do {                                                                                               \
    if (g_spatten_trace && threadIdx.x == (thread))                                                  \
      g_spatten_trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * SPATTEN_TRACE_SLOTS + (slot)] = \
          __builtin_readcyclecounter();                                                              \
  } while (0)
#define SPATTEN_TSTAMP(slot) SPATTEN_TSTAMP_T(slot, 0)
#else
#define SPATTEN_TSTAMP(slot)
#define SPATTEN_TSTAMP_T(slot, thread)
#endif

#ifdef SPATTEN_CHAIN_TRACE   // developer instrumentation of the chained launch (tools/mb/chain_trace.py): per (layer, workgroup)
                             // phase stamps on the device-wide 100 MHz clock
static __device__ unsigned long long* g_chain_trace = nullptr;
#define SPATTEN_CSTAMP(slot)                                                                                            \
  do {                                                                                                                  \
    if (CHAIN && g_chain_trace && threadIdx.x == 0)                                                                     \
      g_chain_trace[((size_t)p.ch_layer * (gridDim.x * gridDim.y * gridDim.z) +                                         \
                     ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (slot)] = wall_clock64(); \
  } while (0)
#define SPATTEN_CSTAMP_L(layer, slot)     /* outside decode_body (the chain loop): the layer is named */                \
  do {                                                                                                                  \
    if (g_chain_trace && threadIdx.x == 0)                                                                              \
      g_chain_trace[((size_t)(layer) * (gridDim.x * gridDim.y * gridDim.z) +                                            \
                     ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define SPATTEN_CSTAMP(slot)
#define SPATTEN_CSTAMP_L(layer, slot)
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
  float* lse; int lse_q;   // (max, sum) per softmax row at [((b*H + h) * lse_q + qi) * 2]; lse_q = rows per (b, h)
  const int32_t* head_ids;   // optional: blockIdx.y -> query head (head pruning: only the kept heads are launched)
  // progressive-quant key planes (KSRC != 0, pq.hip): 4-bit MSB / LSB planes [B,Hkv,cap,D/2] + per-row scale
  const uint8_t* pq_msb; const uint8_t* pq_lsb; const float* pq_scale; int64_t pl_sb, pl_sh, ps_sb, ps_sh;
  float pq_thr; int32_t* pq_need;   // [B*H]: written by the MSB pass (max prob < thr), read by the refetch pass
  // cascade importance (CASC): the previous step's stash + (max, sum) folded into acc while this step's keys stream
  const T* prev_scores; int64_t pv_sb, pv_sh; const float* prev_lse; float* acc; int64_t acc_sh; int prev_len;
  float* head_abs;      // optional [B*H]: += sum_e |out[b, h, e]| (head importance, README.md:21), by the unit's last writer
  const int32_t* step;  // DYN: the device-resident step state (step.hip): word 0 = the cache length AFTER this step's append.
                        // N below is then only a BOUND (grid / chunk / load addresses come from it); cos / sin point at the
                        // state's two staged rotary rows (row 0: the query's position, row 1: the appended key's slot)
  int nr_row;           // rotary-table row of the appended key's slot (N - 1; DYN: 1)
  unsigned long long* ws_part;   // [units][ws_unit] {value, tag} granules; split s of a unit at s * (D + 2)
  unsigned* ws_cnt;     // [units][2]: {arrival counter, launch generation}
  unsigned* ws_err;     // device error word (workspace header)
  int64_t ws_unit;      // granules reserved per unit = ws_splits * (D + 2): FIXED per workspace, so launches with
                        // different split counts / head subsets / key sources never alias another unit's partials
  int B, H, Hkv, N, pos_q, S, chunk, n_q, causal, vis0, append, poll_merge;   // causal: query row qi sees keys [0, vis0 + qi)
  float sqrt_d;
  // FUSED (decode_qkv_kernel): the step's q / k / v projections are computed by the launch itself — x [hidden] (the layer's
  // input row), wqkv [3 * H * D, hidden] the stacked projection weight (rows: all q heads, all k heads, all v heads; row
  // stride w_sn), optional bias [3 * H * D]; xch [units][3 * D] {value, tag} granules: the exchange of a head's q / k / v
  // elements between its splits
  const T* x; const T* wqkv; int64_t w_sn; const T* qkv_bias; unsigned long long* xch; int hidden;
  // ... and (OPROJ) the step's OUTPUT projection (modify_llama.py:163): ow [n_out, H * D] row stride ow_sn, optional bias, y [n_out];
  // ych [B * H * D] granules: the merged attention outputs of all heads, gathered by every workgroup; ych_gen: that
  // exchange's own generation word (the per-head generations above may differ between heads)
  const T* ow; int64_t ow_sn; const T* o_bias; T* y; unsigned long long* ych; unsigned* ych_gen; int n_out;
  // CHAIN (decode_chain.hip): this launch serves SEVERAL layers of one token.  ch_wait [ch_wait_n] = the completion words of the
  // layer this one depends on (NULL for the first layer): the workgroup issues its K/V tile, THEN waits until every word equals
  // ch_tag, THEN loads the query and the appended token (they are functions of the previous layer's output,
  // modify_llama.py:72-92); ch_done = this layer's completion words (one per launched (b, head)): the unit's merger stores
  // ch_tag at [b * ch_ny + blockIdx.y] after its `out` row (ch_ny = heads this layer launches)
  const unsigned* ch_wait; int ch_wait_n; unsigned* ch_done; int ch_ny; unsigned ch_tag;
  int ch_layer;         // (trace only)
  int ch_h;             // the head this workgroup column serves in this layer (the chain loop resolves the layer's head list)
  unsigned* ch_hdr;     // non-NULL on the LAST launched layer: the chain workspace header — word 1 = the token epoch (ch_tag - 1),
                        // word 2 = a counter of that layer's completed units; the unit that completes the layer advances the epoch
};

#ifndef SPATTEN_PQ_UP
#define SPATTEN_PQ_UP 4          // row-groups per pipelined tile of the MSB-plane pass (A/B switch, see launch_decode)
#endif
constexpr int kDecodeThreads = 256;
constexpr int kGemvChunksFused = 8;      // = gemv.hip's kGemvChunks: the fused projection keeps its summation order
// (co-residency bound of the polling merge: common.h coresident_workgroups() — the device's CU count, asked per device)

// one 8-byte {value, tag} granule of a published partial (tag != 0 <=> the value has landed)
__device__ inline void store_granule(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// CHAIN: a unit (b, launched head) of a layer is complete — its `out` row is stored: publish the completion word; on the last
// launched layer also count the unit, and the one that completes the layer re-arms the counter and advances the token epoch
// (every workgroup read the epoch before it could get here: completing the last layer needs the work of all of them).
template <typename T>
__device__ inline void chain_complete(const DecodeParams<T>& p, int b) {
  __hip_atomic_store(p.ch_done + b * p.ch_ny + (int)blockIdx.y, p.ch_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.ch_hdr != nullptr) {
    const unsigned old = __hip_atomic_fetch_add(p.ch_hdr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == (unsigned)(p.B * p.ch_ny)) {
      __hip_atomic_store(p.ch_hdr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.ch_hdr + 1, p.ch_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One tile of a split: UNR row-groups of keys and values in flight / in registers per lane (see decode_body).
template <typename T, int UNR>
struct DecodeTile {
using raw_t = typename Vec8<T>::raw;
  raw_t k_lo[UNR], k_hi[UNR], v_lo[UNR], v_hi[UNR];
  // PQ: a plane row is D/2 bytes and is fetched as 16-byte pieces — LPP = D/32 lanes per row, so ONE wave instruction
  // covers the wave's rows of TWO row-groups (u, u+1): lanes [0,32) hold group u, lanes [32,64) group u+1 (r03: the
  // 4-byte pieces of r02 used a quarter of a line per lane and streamed at 4.2 TB/s where the 16-bit keys reach 5.2)
  u32x4 pm[(UNR + 1) / 2], pl[(UNR + 1) / 2];
  float pscale[(UNR + 1) / 2];
  T prev[UNR];                                               // CASC: the previous step's logit of the row
  T prev2[UNR];                                              // CASC && DYN: the same from the OTHER stash buffer (the device
                                                             // step count decides which of the two is "previous")
};

// MODE 0: the fused decode step.  MODE 1 (scores only): stash + (max, sum), no V traffic, no output — first pass of
// local V pruning.  Compile-time so the hot instantiation carries no extra branches.
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
// CASC: cascade (cumulative) importance, deferred by one step — see DecodeParams.
// DYN: the cache length lives in DEVICE memory (DecodeParams::step) so that ONE captured HIP graph of a decode step
// replays for every token of a turn.  Everything an ADDRESS of a tile load depends on stays a launch constant — the
// split's first row, the static chunk, the bound p.N — and the loaded length only gates the arithmetic (which rows are
// live, where the appended row goes): the loads are issued exactly as in the static kernel, nothing waits for the
// length.  Rows [length, bound) of the planes are read and discarded (weight 0), so they must hold finite values
// (include/spatten.h: zero-fill the planes once).
// FUSED: the attention half of decode_qkv_kernel (below): the query and the appended token's key / value do not come from
// memory but from LDS (`s_x`: q | k | v of this head, 3 * D floats holding model-dtype values), filled by the projection
// waves of the same workgroup; two workgroup barriers are added — B1 in front of the tile loads (the projection waves have
// issued their last weight pass: the K/V stream follows the weight stream through the memory pipe), B2 behind them (the
// projections of the whole head have been exchanged).  The rest — tile arithmetic, reduction, publication, merge — is the
// plain step's, bit for bit.
template <typename T, int D, int UNR, int MODE = 0, bool LEAN = false, int KSRC = 0, bool NT = LEAN, bool CASC = false,
          bool PIPE = false, bool DYN = false, bool FUSED = false, bool OPROJ = false, int THREADS = kDecodeThreads,
          bool HIDS = false, int CHAIN = 0>
__device__ __forceinline__ void decode_body(const DecodeParams<T>& p, const float* s_x = nullptr, T* s_ch = nullptr) {
  constexpr bool SCORES_ONLY = (MODE == 1);
  constexpr bool PQ = (KSRC != 0);
  constexpr int LPR = D / 16;                    // lanes per row
  constexpr int RPI = THREADS / LPR;             // rows per row-group (one per thread group of LPR lanes)
  constexpr int NW = THREADS / kWave;            // waves of the attention team (4; 8 in the two-waves-per-SIMD form)
  constexpr int TILE = RPI * UNR;
  constexpr int HALF = D / 2;
  constexpr int G = (THREADS / D) > 0 ? (THREADS / D) : 1;   // merge thread groups
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  using D8 = Dot8<T>;

  __shared__ float s_o[NW][D + 2];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x;
  const int c = tid % LPR;
  const int r = tid / LPR;
  const int wave = tid / kWave;
  const int lane = tid % kWave;
  SPATTEN_TSTAMP(0);

  // grid = (S, H, B * n_q): no integer divisions on the way to the first load
  const int split = blockIdx.x;
  // HIDS: the lean step over a LIST of heads (head pruning / a head-parallel rank's survivors): one scalar load on the way to
  // the first tile load instead of the general kernel's whole argument block
  const int h = CHAIN ? p.ch_h : HIDS ? p.head_ids[blockIdx.y] : ((!LEAN && p.head_ids) ? p.head_ids[blockIdx.y] : (int)blockIdx.y);
  const int b = CHAIN ? (p.B == 1 ? 0 : (int)blockIdx.z % p.B) : (LEAN || p.n_q == 1) ? (int)blockIdx.z : (int)blockIdx.z / p.n_q;
  const int qi = (LEAN || p.n_q == 1) ? 0 : (int)blockIdx.z - b * p.n_q;
  const int hkv = (LEAN || p.Hkv == p.H) ? h : h / (p.H / p.Hkv);
  const int unit = LEAN ? (b * p.H + h) : (b * p.H + h) * p.n_q + qi;   // one softmax row
  if (KSRC == 2 && p.pq_need[unit] == 0) return;   // confident head: the MSB pass already produced its output

  // DYN: the length is requested FIRST (vector loads return in order: it is back before the query) and read after the
  // tile loads have been issued
  int n_dyn = 0, cnt_dyn = 0, pl_dyn = 0;
  if (DYN) n_dyn = p.step[opaque_lane(0)];
  if (DYN && (CASC || KSRC == 2)) { cnt_dyn = p.step[2 + opaque_lane(0)]; pl_dyn = p.step[3 + opaque_lane(0)]; }
  // CASC: (max, sum) of the previous step's row, requested BEFORE the tile (r03: behind the tile these 8 bytes came back
  // after it — returns are in order — and the first accumulation waited for the whole stream); DYN: of both buffers
  float ml_a[2] = {0.f, 1.f}, ml_b[2] = {0.f, 1.f};
  if (CASC) {
    const float* ml = p.prev_lse + 2 * (b * p.H + h) + opaque_lane(0);
    ml_a[0] = ml[0]; ml_a[1] = ml[1];
    if (DYN) { const float* m2 = p.lse + 2 * (b * p.H + h) + opaque_lane(0); ml_b[0] = m2[0]; ml_b[1] = m2[1]; }
  }
  // rows [lo, lo + chunk) of this split: chunks are balanced (ceil(N / S), any N) and need not be whole tiles — the last
  // tile of a chunk runs with fewer live row-groups.  The tile loads touch rows [lo, rl): a launch constant.
  const int lo = split * p.chunk;
  int rl;
  {
    const int n_vis_s = (!LEAN && p.causal) ? min(p.N, p.vis0 + qi) : p.N;
    const int hi_all_s = min(lo + p.chunk, (p.scores != nullptr) ? p.N : n_vis_s);
    rl = (!DYN && p.append && lo < p.N && hi_all_s == p.N) ? hi_all_s - 1 : hi_all_s;
  }

  T* krbase = p.krc + b * p.kv_sb + hkv * p.kv_sh;
  T* vbase = p.vc + b * p.kv_sb + hkv * p.kv_sh;

  // ---- every load of the first tile goes out before anything is waited for -----------------------------------------
  // A tile = UNR row-groups (16-bit dtypes at d = 128: 10 x 32 = 320 rows, 160 registers of K/V in flight per lane):
  // at Llama-2-7B decode sizes a split's whole chunk (N / 8 <= 320 rows) is ONE tile.  The memory system is saturated
  // by the other 255 workgroups, so each dependent request -> wait -> compute round costs ~5k cycles of queueing delay
  // whatever its size (a chunk of 264 rows run as 128 + 128 + 8 measured 15.8k cycles of streaming against 9.4k for
  // 2 x 128): rounds, not bytes, are what a small launch must minimise.
  // Vector loads of a wave return IN ORDER and the compiler only places an exact `s_waitcnt vmcnt(N)` in front of a use
  // when no control flow surrounds the loads, so the issue order below IS the schedule: query + its rotary row, the
  // keys, the new token's key, the values, the new token's value — all unconditional (rows past the chunk's end
  // re-read its last row: cache hits; a launch that appends nothing reads the query in place of the new token) — and
  // then the arithmetic in the same order: rotate the query, score row-group u while u+1.. are still in flight,
  // softmax, P·V of value group u while u+1.. are still in flight.  (With the loads under `if (u < ng)` the compiler
  // fell back to vmcnt(0) after the last issue and ALL arithmetic ran after the stream: 9.0 us for a kernel whose
  // stream alone takes 6.5.)
  using Tile = DecodeTile<T, UNR>;
  Tile tile_a;
  Tile tile_b;   // PIPE only: the second half of the double buffer (dead code otherwise)
  const uint8_t* pq_m = PQ ? p.pq_msb + b * p.pl_sb + hkv * p.pl_sh : nullptr;
  const uint8_t* pq_l = KSRC == 2 ? p.pq_lsb + b * p.pl_sb + hkv * p.pl_sh : nullptr;
  const float* pq_s = PQ ? p.pq_scale + b * p.ps_sb + hkv * p.ps_sh : nullptr;
  const T* prevp = CASC ? p.prev_scores + b * p.pv_sb + h * p.pv_sh : nullptr;
  const T* prevp2 = (CASC && DYN) ? p.scores + b * p.sc_sb + h * p.sc_sh : nullptr;
  const int prev_clamp = (CASC && DYN) ? p.N : p.prev_len;     // static bound of the previous-row addresses
  auto row_of = [&](int t0, int u, int rlim = -1) { return max(min(t0 + u * RPI + r, (rlim < 0 ? rl : rlim) - 1), 0); };   // (an empty split reads row 0)
  // PQ lane mapping (see Tile): lane = (g * RW + r8) * LPP + cc — row-group u + g, row r8 of the wave's RW rows, 16-byte piece cc
  constexpr int LPP = D / 32, RW = kWave / LPR;
  const int pq_cc = lane % LPP, pq_rr = lane / LPP, pq_g = pq_rr / RW, pq_r8 = pq_rr % RW;
  auto issue_keys = [&](Tile& tl, int t0, const T* krb = nullptr, int rlim = -1) {
    if (krb == nullptr) krb = krbase;
    if (PQ) {
#pragma unroll
      for (int u = 0; u < UNR; u += 2) {
        const int j = max(min(t0 + (u + pq_g) * RPI + wave * RW + pq_r8, rl - 1), 0);
        const int64_t po = (int64_t)j * HALF + 16 * pq_cc;   // a plane row is D/2 bytes
        tl.pm[u / 2] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pq_m + po));
        if (KSRC == 2) tl.pl[u / 2] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pq_l + po));
        tl.pscale[u / 2] = pq_s[j];
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = row_of(t0, u, rlim);
      if (PQ) {
      } else {
        const T* kp = krb + (int64_t)j * D;
        tl.k_lo[u] = NT ? V8::ldg_stream(kp + 8 * c) : V8::ldg(kp + 8 * c);
        tl.k_hi[u] = NT ? V8::ldg_stream(kp + HALF + 8 * c) : V8::ldg(kp + HALF + 8 * c);
      }
      if (CASC) tl.prev[u] = prevp[max(min(j, prev_clamp - 1), 0)];
      if (CASC && DYN) tl.prev2[u] = prevp2[max(min(j, prev_clamp - 1), 0)];
    }
  };
  auto issue_values = [&](Tile& tl, int t0, const T* vb = nullptr, int rlim = -1) {
    if (vb == nullptr) vb = vbase;
    if (!SCORES_ONLY) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const T* vp = vb + (int64_t)row_of(t0, u, rlim) * D;
        tl.v_lo[u] = NT ? V8::ldg_stream(vp + 8 * c) : V8::ldg(vp + 8 * c);
        tl.v_hi[u] = NT ? V8::ldg_stream(vp + HALF + 8 * c) : V8::ldg(vp + HALF + 8 * c);
      }
    }
  };
  raw_t q_raw[4], n_raw[2], nk_raw[2], nv_raw[2];
  unsigned gen_c = 0;
  if (CHAIN) {
    // The K/V tile depends on nothing upstream: it goes out BEFORE the dependency of this layer on the previous one is waited
    // for — by waves 1.. of the workgroup at once.  Wave 0 first polls the previous layer's completion words, then fetches the
    // head's q | k_new | v_new rows (3 x 16 lanes x 16 bytes: ONE load instruction) into the LDS staging rows s_ch[0..3), and
    // only then requests ITS rows of the tile: a wave's loads return in order, so a poll behind 20 KB of tile requests would
    // see the flag when the tile has landed — with the stream bandwidth-bound, a whole stream late (measured: the layer period
    // stayed at the per-layer launch's, tools/mb/chain_trace.py).  The other waves wait at a bare s_barrier (no vmcnt drain:
    // their tiles stay in flight) and read what they need from LDS.  Rows 3 / 4 of s_ch (the rotary rows of the query's position
    // and of the appended slot) are the same for every layer: staged once by the chain loop.
    // The same order serves the merge: the unit's merger (waves 0-1 of its last split) re-enters here only after the merge, so
    // its polls for the partials never queue behind the next layer's tile either.
    // (Measured and dropped: the MERGING workgroup — which comes out of the previous layer's merge with the completion words
    //  set — requesting its whole tile behind the barrier, so that wave 0's poll does not queue behind the other waves' 140 KB in
    //  the CU's one vector-memory queue: 9.9 us per layer against 9.3, its stream then starts 1 us later still.  And, late in
    //  round 6: only the merger's first poll ISSUED in front of them — one bare barrier more in that workgroup: 10.5-10.6 against
    //  9.35-9.4; the other splits' granules requested before the merger's own reduction: 9.65 against 9.35; the merger folding its
    //  own partial from registers instead of publishing it and reading it back: 9.5-9.55 against 9.36-9.44.  HISTORY.md §0.)
    const bool w0 = __builtin_amdgcn_readfirstlane(wave) == 0;
    const bool late = w0;
    if (!late) {
      issue_keys(tile_a, lo);
      issue_values(tile_a, lo);
    }
    gen_c = p.ws_cnt[(p.S > 1 ? 2 * unit + 1 : 0) + opaque_lane(0)];
    SPATTEN_CSTAMP(0);
    if (w0) {
      if (p.ch_wait != nullptr) {
        int spins = 0;
        bool ok;
        do {
          unsigned bad = 0u;
          for (int u = lane; u < p.ch_wait_n; u += kWave)
            bad |= __hip_atomic_load(p.ch_wait + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ p.ch_tag;
          ok = __all(bad == 0u);
          if (!ok) __builtin_amdgcn_s_sleep(1);
        } while (!ok && ++spins < (1 << 20));
        if (!ok && lane == 0) atomicOr(p.ws_err, 2u);      // (bounded: the launch ends, the status call reports a timeout)
      }
      SPATTEN_CSTAMP(1);
      constexpr int PPR = D / 8;                           // 16-byte pieces per row
      if (lane < 3 * PPR) {
        const int row = lane / PPR, pc = lane % PPR;
        const T* src = row == 0 ? p.q + b * p.q_sb + h * p.q_sh
                                : (row == 1 ? p.k_new : p.v_new) + b * p.new_sb + hkv * p.new_sh;
        V8::stg(s_ch + row * D + 8 * pc, V8::ldg(src + 8 * pc));
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    SPATTEN_CSTAMP(2);
    if (late) {      // (behind the barrier: the other waves start on their rows while these 20 requests pass the texture path)
      issue_keys(tile_a, lo);
      issue_values(tile_a, lo);
    }
    q_raw[0] = V8::ldg(s_ch + 8 * c);
    q_raw[1] = V8::ldg(s_ch + HALF + 8 * c);
    q_raw[2] = V8::ldg(s_ch + 3 * D + 8 * c);
    q_raw[3] = V8::ldg(s_ch + 3 * D + HALF + 8 * c);
  } else
  {
    const T* qp = p.q + b * p.q_sb + h * p.q_sh + (LEAN ? 0 : qi * p.q_sq);
    int pq = (!LEAN && p.pos_ids) ? (int)p.pos_ids[b * p.pos_sb + qi] : p.pos_q + qi;
    pq = min(max(pq, 0), p.table_rows - 1);
    if (!FUSED) {
      q_raw[0] = V8::ldg(qp + 8 * c);
      q_raw[1] = V8::ldg(qp + HALF + 8 * c);
    }
    q_raw[2] = V8::ldg(p.cos + (int64_t)pq * HALF + 8 * c);
    q_raw[3] = V8::ldg(p.sin + (int64_t)pq * HALF + 8 * c);
  }
  unsigned gen_f = 0, gen_y = 0;
  if (FUSED) {      // what does not depend on the projections goes out first; then B1: the projection waves are on their last pass
    n_raw[0] = V8::ldg(p.cos + (int64_t)p.nr_row * HALF + 8 * c);
    n_raw[1] = V8::ldg(p.sin + (int64_t)p.nr_row * HALF + 8 * c);
    gen_f = p.ws_cnt[(p.S > 1 ? 2 * unit + 1 : 0) + opaque_lane(0)];
    if (OPROJ) gen_y = p.ych_gen[opaque_lane(0)];
#if !(defined(SPATTEN_FUSED_EXP) && SPATTEN_FUSED_EXP == 2)
    // B1 — a bare s_barrier: __syncthreads() carries a fence, i.e. `s_waitcnt vmcnt(0)`, and the projection waves would
    // drain their in-flight weight passes in front of it (r04: the fused launch then took 29.5 us, the two launches 29.0)
    __builtin_amdgcn_s_barrier();
#endif
  }
  if (!CHAIN) issue_keys(tile_a, lo);
  if (!FUSED && !CHAIN) {   // the new token's un-rotated key and the rotary row of its slot N-1 (modify_llama.py:103-104)
    const T* kp = p.k_new + b * p.new_sb + hkv * p.new_sh;      // (always readable: see `append`)
    nk_raw[0] = V8::ldg(kp + 8 * c);
    nk_raw[1] = V8::ldg(kp + HALF + 8 * c);
    n_raw[0] = V8::ldg(p.cos + (int64_t)p.nr_row * HALF + 8 * c);
    n_raw[1] = V8::ldg(p.sin + (int64_t)p.nr_row * HALF + 8 * c);
  }
  if (!CHAIN) issue_values(tile_a, lo);
  if (!FUSED && !CHAIN) {
    const T* vp = p.v_new + b * p.new_sb + hkv * p.new_sh;
    nv_raw[0] = V8::ldg(vp + 8 * c);
    nv_raw[1] = V8::ldg(vp + HALF + 8 * c);
  }
  // this unit's launch generation (tags of the published partials, see below): written by the previous launch's merger,
  // so it comes from memory — a VECTOR load queued behind the tile (a scalar load would be waited for with the kernel
  // arguments, before anything else happens).  (No workspace: ws_cnt points at the rotary table.)
  const unsigned gen = FUSED ? gen_f : CHAIN ? gen_c : p.ws_cnt[(p.S > 1 ? 2 * unit + 1 : 0) + opaque_lane(0)];
  // nothing that consumes a load may be scheduled above this point: left alone, the scheduler hoists the query
  // rotation (and the wait for the query) in front of the tile loads, which then leave one memory latency late
  __builtin_amdgcn_sched_barrier(0);
  if (FUSED) {
#if defined(SPATTEN_FUSED_EXP) && SPATTEN_FUSED_EXP == 2      // A/B harness only: the tile goes out at kernel start
    __builtin_amdgcn_s_barrier();
#endif
    // B2: s_x holds the head's q | k | v (the writing wave drained its LDS stores in front of its s_barrier).  Bare again: the
    // tile loads of this wave stay in flight across it and are waited for one row-group at a time below
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = s_x[D + 8 * c + i];
    nk_raw[0] = V8::pack(t8);
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = s_x[D + HALF + 8 * c + i];
    nk_raw[1] = V8::pack(t8);
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = s_x[2 * D + 8 * c + i];
    nv_raw[0] = V8::pack(t8);
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = s_x[2 * D + HALF + 8 * c + i];
    nv_raw[1] = V8::pack(t8);
  }
  SPATTEN_TSTAMP(5);
  // ---- the live rows of this split ------------------------------------------------------------------------------
  const int N = DYN ? __builtin_amdgcn_readfirstlane(n_dyn) : p.N;
  // CASC && DYN: the two stash / (max, sum) buffers swap roles every step — step k (1-based count in the state) writes
  // buffer (k - 1) & 1 and folds the other one, whose rows [0, prev_len) the previous step wrote (0 after a set)
  // (the refetch pass of progressive quantisation follows a CASC pass 1 of the same step: it writes the same buffer)
  const bool odd = (DYN && (CASC || (KSRC == 2 && p.prev_scores != nullptr))) ? ((__builtin_amdgcn_readfirstlane(cnt_dyn) - 1) & 1) != 0 : false;
  const int prev_len = (CASC && DYN) ? __builtin_amdgcn_readfirstlane(pl_dyn) : p.prev_len;
  // keys this query may attend to (HF causal: j <= P + i with P = N - n_q); the stash covers all N
  const int n_vis = (!LEAN && p.causal) ? min(N, p.vis0 + qi) : N;
  const int hi_all = min(lo + p.chunk, (p.scores != nullptr) ? N : n_vis);
  // The split that ends at N appends the new token (row N-1): its K/V rows come from k_new / v_new, not from the
  // cache, so it is scored as ONE EXTRA ROW beside the tile and the tile covers the rows already cached, [lo, hi).
  const bool owns_new = p.append && lo < N && hi_all == N;
  const int hi = owns_new ? hi_all - 1 : hi_all;
  auto groups_of = [&](int t0) { return max(0, min(UNR, (hi - t0 + RPI - 1) / RPI)); };

  // ---- rotate the query (its data was requested first, so it is here long before the keys) -----------------------
  typename D8::packed q_lo, q_hi;                // rotated query, packed in the model dtype (exact: it IS rounded)
  typename NibbleDot<T>::packed qn[4];           // PQ: the same rotated query, arranged for the nibble dot product
  {
    float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
    if (FUSED) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { xlo[i] = s_x[8 * c + i]; xhi[i] = s_x[HALF + 8 * c + i]; }
    } else {
      V8::unpack(q_raw[0], xlo);
      V8::unpack(q_raw[1], xhi);
    }
    V8::unpack(q_raw[2], cc);
    V8::unpack(q_raw[3], ss);
    rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
    q_lo = D8::pack(ylo);
    q_hi = D8::pack(yhi);
    if (PQ) {
      // the rotated query re-dealt for the plane mapping: piece cc covers elements [32 cc, 32 cc + 32) = 4 dwords of 8
      // nibbles; dword k's elements sit in lane c_src of this lane's LPR-group, in its lower or upper half-row
      const int half = pq_cc / (LPP / 2), grp0 = lane & ~(LPR - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int src = grp0 + (pq_cc % (LPP / 2)) * 4 + k;
        float e8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float lo_ = __shfl(ylo[i], src, kWave), hi_ = __shfl(yhi[i], src, kWave);
          e8[i] = half ? hi_ : lo_;
        }
        qn[k] = NibbleDot<T>::prep(e8);
      }
    }
  }
  SPATTEN_TSTAMP(6);
  SPATTEN_CSTAMP(3);
  const float rsqrt_d = 1.0f / p.sqrt_d;
  const T* maskp = (!LEAN && p.mask) ? p.mask + b * p.mask_sb + qi * p.mask_sq : nullptr;
  T* stashp = p.scores ? p.scores + b * p.sc_sb + h * p.sc_sh + (LEAN ? 0 : qi * p.sc_sq) : nullptr;
  if (DYN && odd) stashp = const_cast<T*>(p.prev_scores) + b * p.pv_sb + h * p.pv_sh;
  float* lse_cur = (DYN && odd) ? const_cast<float*>(p.prev_lse) : p.lse;
  float casc_m = 0.f, casc_rl = 0.f;
  float* accp = nullptr;
  if (CASC) {
    const float pm = odd ? ml_b[0] : ml_a[0], pl = odd ? ml_b[1] : ml_a[1];
    casc_m = (pm == -INFINITY) ? 0.f : pm;
    casc_rl = 1.0f / pl;
    accp = p.acc + h * p.acc_sh;
  }

  // per-THREAD online softmax (the LPR lanes of a row share its score, so they agree): no barrier and
  // no cross-lane maximum inside the loop; the row groups are reconciled once, after the loop.
  float m_run = -INFINITY, l_run = 0.f;
  float olo[8], ohi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = 0.f; ohi[i] = 0.f; }

  // with_new (wave-uniform): this is the tile after which the owning split also scores the appended row
  auto process_tile = [&](Tile& tl, int t0, int ng, bool with_new) {
    float mk[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) mk[u] = (maskp && u < ng) ? DT<T>::to_f32(maskp[min(t0 + u * RPI + r, n_vis - 1)]) : 0.f;

    // ---- scores of the row groups, in the order their keys arrive (the values are still in flight) ------------
    float sc[UNR];
    if (PQ) {
      // sum_i q_i * q8_i with q8 = 16 * sext(msb nibble) + lsb nibble over this lane's 32 elements, reduced over the LPP
      // lanes of the row, scaled per row; then dealt back to the value mapping: lane (row r, any c) of group u takes the
      // logit from plane lane (g = u & 1, r8 = r, cc = 0)
#pragma unroll
      for (int u = 0; u < UNR; u += 2) {
        float a = 0.f;
        if (u < ng) {
#pragma unroll
          for (int k = 0; k < 4; ++k) a += NibbleDot<T>::dot(qn[k], tl.pm[u / 2][k] ^ 0x88888888u, 8.f);
          a *= 16.f;
          if (KSRC == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a += NibbleDot<T>::dot(qn[k], tl.pl[u / 2][k], 0.f);
          }
          a += dpp_mov<kDppXor1>(a);
          if (LPP == 4) a += dpp_mov<kDppXor2>(a);
          a = a * tl.pscale[u / 2] / p.sqrt_d;                  // fp32 logits
        }
        const int r8v = (lane / LPR) * LPP;
        sc[u] = __shfl(a, r8v, kWave);
        if (u + 1 < UNR) sc[u + 1] = __shfl(a, RW * LPP + r8v, kWave);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!PQ) sc[u] = 0.f;
      if (u < ng) {
        if (PQ) {
        } else {
          const float a = group_sum<LPR>(D8::dot(q_hi, tl.k_hi[u], D8::dot(q_lo, tl.k_lo[u], 0.f)));
          // matmul result -> dtype, then the separate divide -> dtype (modify_llama.py:111-113)
          sc[u] = DT<T>::round(div_by_const(DT<T>::round(a), p.sqrt_d, rsqrt_d));
        }
      }
    }
    // ---- the appended token (owning split, after its last tile): K un-rotated into the cache (modify_llama.py:95-100),
    // its rotation into the shadow, V — stored by the lanes of row-group slot 0 — and its logit, one extra row that
    // those lanes fold into their softmax
    float s_new = -INFINITY;
    if (with_new) {
      if (CHAIN) {     // the appended token's key and the rotary row of its slot: staged in LDS (see above)
        nk_raw[0] = V8::ldg(s_ch + D + 8 * c);
        nk_raw[1] = V8::ldg(s_ch + D + HALF + 8 * c);
        n_raw[0] = V8::ldg(s_ch + 4 * D + 8 * c);
        n_raw[1] = V8::ldg(s_ch + 4 * D + HALF + 8 * c);
      }
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      V8::unpack(nk_raw[0], xlo);
      V8::unpack(nk_raw[1], xhi);
      V8::unpack(n_raw[0], cc);
      V8::unpack(n_raw[1], ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
      const raw_t r_lo = V8::pack(ylo), r_hi = V8::pack(yhi);
      const int jn = N - 1;
      if (r == 0) {
        T* kbase = p.kc ? p.kc + b * p.kv_sb + hkv * p.kv_sh : nullptr;
        if (kbase) {
          V8::stg(kbase + (int64_t)jn * D + 8 * c, nk_raw[0]);
          V8::stg(kbase + (int64_t)jn * D + HALF + 8 * c, nk_raw[1]);
        }
        V8::stg(krbase + (int64_t)jn * D + 8 * c, r_lo);
        V8::stg(krbase + (int64_t)jn * D + HALF + 8 * c, r_hi);
      }
      const float a = group_sum<LPR>(D8::dot(q_hi, D8::pack(yhi), D8::dot(q_lo, D8::pack(ylo), 0.f)));
      float s = DT<T>::round(div_by_const(DT<T>::round(a), p.sqrt_d, rsqrt_d));
      if (r == 0) {
        if (stashp != nullptr && c == 0) stashp[jn] = DT<T>::from_f32(s);              // pre-mask (:116-119)
        if (maskp != nullptr) s = DT<T>::round(s + DT<T>::to_f32(maskp[jn]));           // :132
        s_new = jn < n_vis ? s : -INFINITY;
      }
    }
#ifdef SPATTEN_TRACE
    SPATTEN_TSTAMP(7);
#endif
    float m_new = fmaxf(m_run, s_new);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = t0 + u * RPI + r;
      const bool valid = u < ng && j < hi;
      float s = sc[u];
      if (stashp != nullptr && c == 0 && valid) stashp[j] = DT<T>::from_f32(s);       // pre-mask (:116-119)
      if (CASC && c == 0 && valid && j < prev_len)                                    // last step's probability of key j
        atomicAdd(accp + j, __expf(DT<T>::to_f32((CASC && DYN && odd) ? tl.prev2[u] : tl.prev[u]) - casc_m) * casc_rl);
      if (maskp != nullptr) s = DT<T>::round(s + mk[u]);                               // :132
      s = (valid && j < n_vis) ? s : -INFINITY;
      sc[u] = s;
      m_new = fmaxf(m_new, s);
    }
    if (m_new > m_run) {                           // first tile; rare afterwards
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
    // ---- P·V, two value rows at a time, in arrival order --------------------------------------------------------
    if (!SCORES_ONLY) {
#pragma unroll
      for (int u = 0; u < UNR; u += 2) {
        if (u + 1 < ng) {
          const typename PairFma<T>::prob2 pp = PairFma<T>::pack_p(pj[u], pj[u + 1]);
          PairFma<T>::fma(olo, tl.v_lo[u], tl.v_lo[u + 1], pp);
          PairFma<T>::fma(ohi, tl.v_hi[u], tl.v_hi[u + 1], pp);
        } else if (u < ng) {                       // odd group count: the last row alone
          const typename PairFma<T>::prob2 pp = PairFma<T>::pack_p(pj[u], 0.f);
          PairFma<T>::fma(olo, tl.v_lo[u], tl.v_lo[u], pp);
          PairFma<T>::fma(ohi, tl.v_hi[u], tl.v_hi[u], pp);
        }
      }
      if (with_new) {                              // the appended row: its V goes to the cache and into the product
        if (CHAIN) { nv_raw[0] = V8::ldg(s_ch + 2 * D + 8 * c); nv_raw[1] = V8::ldg(s_ch + 2 * D + HALF + 8 * c); }
        const float pn = (s_new == -INFINITY) ? 0.f : __expf(s_new - m_run);
        l_run += pn;
        const typename PairFma<T>::prob2 pp = PairFma<T>::pack_p(pn, 0.f);
        PairFma<T>::fma(olo, nv_raw[0], nv_raw[0], pp);
        PairFma<T>::fma(ohi, nv_raw[1], nv_raw[1], pp);
        if (r == 0) {
          V8::stg(vbase + (int64_t)(N - 1) * D + 8 * c, nv_raw[0]);
          V8::stg(vbase + (int64_t)(N - 1) * D + HALF + 8 * c, nv_raw[1]);
        }
      }
    }
  };
  // the tile after which the owning split scores the appended row: the one that holds its last cached row
  const int last_row = max(hi - 1, lo);
  auto with_new = [&](int t0) { return owns_new && t0 <= last_row && last_row < t0 + TILE; };
  if (!PIPE) {
    process_tile(tile_a, lo, groups_of(lo), with_new(lo));   // straight-line from the loads to their uses
    for (int t0 = lo + TILE; t0 < hi; t0 += TILE) {          // (longer chunks normally run the PIPE instantiation)
      issue_keys(tile_a, t0);
      issue_values(tile_a, t0);
      __builtin_amdgcn_sched_barrier(0);
      process_tile(tile_a, t0, groups_of(t0), with_new(t0));
    }
  } else {
    // Long chunks (N / S above one single-shot tile): smaller tiles, double buffered — the next tile's loads are in
    // flight while this one is scored and multiplied (in-order returns: waiting for tile A never waits for tile B
    // behind it), so the memory pipe never drains between tiles.  Tiles past the end re-read the last row (cache
    // hits) and process nothing.
    for (int t0 = lo; t0 < hi; t0 += 2 * TILE) {
      issue_keys(tile_b, t0 + TILE);
      issue_values(tile_b, t0 + TILE);
      __builtin_amdgcn_sched_barrier(0);
      process_tile(tile_a, t0, groups_of(t0), with_new(t0));
      issue_keys(tile_a, t0 + 2 * TILE);
      issue_values(tile_a, t0 + 2 * TILE);
      __builtin_amdgcn_sched_barrier(0);
      process_tile(tile_b, t0 + TILE, groups_of(t0 + TILE), with_new(t0 + TILE));
    }
    // the appended token is the ONLY row of its split (N = lo + 1: the loop above ran zero times): score and store it
    if (owns_new && hi <= lo) process_tile(tile_a, lo, 0, true);
  }

#ifdef SPATTEN_EXP_NOREDUCE   // A/B harness only: what does everything after the streaming loop cost?
  {   // (every accumulator stays live: with only olo[0] written the compiler drops the upper-half value loads)
    float keep = l_run + m_run;
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += olo[i] + ohi[i];
    if (tid < D) p.out[b * p.out_sb + h * D + tid] = DT<T>::from_f32(keep);
  }
  return;
#endif
  SPATTEN_TSTAMP(1);
  SPATTEN_CSTAMP(4);
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
  if (CHAIN) {   // a BARE barrier: __syncthreads() carries a fence = s_waitcnt vmcnt(0), i.e. it would wait for the next layer's tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
  float o_tot = 0.f, l_tot = 0.f;
  {
    if constexpr (NW == 4) {
      const float m0 = s_o[0][D + 1], m1 = s_o[1][D + 1], m2 = s_o[2][D + 1], m3 = s_o[3][D + 1];
      const float m_wg = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      const float mu = (m_wg == -INFINITY) ? 0.f : m_wg;
      const float w0 = __expf(m0 - mu), w1 = __expf(m1 - mu), w2 = __expf(m2 - mu), w3 = __expf(m3 - mu);   // exp(-inf) = 0
      if (tid < D) o_tot = (s_o[0][tid] * w0 + s_o[1][tid] * w1) + (s_o[2][tid] * w2 + s_o[3][tid] * w3);
      l_tot = (s_o[0][D] * w0 + s_o[1][D] * w1) + (s_o[2][D] * w2 + s_o[3][D] * w3);
      m_run = m_wg;
    } else {
      float mw[NW], m_wg = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) { mw[w] = s_o[w][D + 1]; m_wg = fmaxf(m_wg, mw[w]); }
      const float mu = (m_wg == -INFINITY) ? 0.f : m_wg;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float ww = __expf(mw[w] - mu);
        if (tid < D) o_tot = fmaf(s_o[w][tid], ww, o_tot);
        l_tot = fmaf(s_o[w][D], ww, l_tot);
      }
      m_run = m_wg;
    }
  }
#ifdef SPATTEN_EXP_NOMERGE    // A/B harness only: what do publish + ticket + merge cost?
  if (tid < D) p.out[b * p.out_sb + h * D + tid] = DT<T>::from_f32(o_tot / l_tot);
  return;
#endif
  SPATTEN_TSTAMP(2);
  SPATTEN_CSTAMP(5);
  T* outp = p.out + b * p.out_sb + (LEAN ? 0 : qi * p.out_sq) + h * D;
  // head importance (head pruning): head_abs[unit] += sum_e |out[e]| of the value this launch leaves in `out` — summed
  // in a fixed order (deterministic); under progressive quantisation the MSB pass adds only for confident heads and
  // the refetch pass for the heads it recomputes.  Called by every thread of the workgroup (uniform condition).
  auto add_head_abs = [&](float val, bool active, bool commit) {
    float v = wave_sum(active ? fabsf(DT<T>::round(val)) : 0.f);
    __syncthreads();
    if (lane == 0) s_o[0][wave] = v;
    __syncthreads();
    if (tid == 0 && commit) {
      float tot = (s_o[0][0] + s_o[0][1]) + (s_o[0][2] + s_o[0][3]);
#pragma unroll
      for (int w = 4; w < NW; ++w) tot += s_o[0][w];
      p.head_abs[unit] += tot;
    }
  };
  if (p.S == 1) {
    if (!SCORES_ONLY && tid < D) outp[tid] = DT<T>::from_f32(o_tot / l_tot);
    if (OPROJ && tid < D) store_granule(p.ych + (int64_t)unit * D + tid, DT<T>::round(o_tot / l_tot), (gen_y & 0x7FFFFFFFu) + 1u);
    if (lse_cur != nullptr && tid == 0) { float* ls = lse_cur + ((int64_t)(b * p.H + h) * p.lse_q + qi) * 2; ls[0] = m_run; ls[1] = l_tot; }
    const bool need1 = KSRC == 1 && (1.0f / l_tot) < p.pq_thr;                          // max prob = exp(0) / sum
    if (KSRC == 1 && tid == 0) p.pq_need[unit] = need1 ? 1 : 0;
    if (!SCORES_ONLY && p.head_abs != nullptr) add_head_abs(o_tot / l_tot, tid < D, !need1);
    if (CHAIN && tid == 0) chain_complete(p, b);
    return;
  }

  // ---- publish the partial; the last split to arrive merges ------------------------------------
  // Every value goes out as ONE 8-byte write-through {value, tag} granule (agent-scope relaxed atomic
  // store = sc1): the data is its own flag, so nobody waits for store acknowledgements.  The ticket
  // tells the last arriver that every other split has ISSUED its granules; it then reads them with
  // agent-scope loads (placement independent across the 8 XCD L2s) and re-reads the rare granule whose
  // tag has not landed yet.  The tag is the unit's launch GENERATION + 1 (a word next to the counter, advanced by
  // the merger): granules of earlier launches never match, so nothing has to be cleared for the next launch
  // (clearing S x (D+2) granules cost the merger 0.3 us of a 13 us kernel).  A unit's region is FIXED
  // (unit * ws_unit, whatever this launch's S): a launch that skips a unit (head list, confident PQ heads) leaves
  // its generation and its region alone, so a later launch can never meet another unit's granules under its own tag.
  unsigned long long* ws = p.ws_part + (int64_t)unit * p.ws_unit;
  unsigned long long* part = ws + (int64_t)split * (D + 2);
  const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;
  // Who merges.  poll_merge (the host sets it when the whole grid is co-resident BY CONSTRUCTION: no more workgroups than
  // CUs, so every workgroup is running or will be started without any other having to finish): the unit's LAST split
  // merges; it simply polls the other splits' granules until their tags match — the hand-off is ONE memory hop, no
  // ticket round trip in front of it.  Otherwise (more workgroups than the chip holds at once: a polling workgroup
  // could wait for one that cannot start) the last ARRIVER merges: a ticket tells it that every split has at least
  // issued its granules, so its polling is bounded by store latency, never by scheduling.
  if (!p.poll_merge) {
    // the ticket is drawn by the LAST wave, which has no stores in flight: on CDNA4 vmcnt also counts stores, so a
    // wave that just issued granules would wait for their write-through acknowledgements before it sees its ticket
    if (tid == THREADS - 1)
      s_ticket = __hip_atomic_fetch_add(p.ws_cnt + 2 * unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < D && tid < THREADS - kWave) store_granule(part + tid, o_tot, tag);
  if (D > THREADS - kWave && tid >= THREADS - kWave && tid < D) store_granule(part + tid, o_tot, tag);   // D = 256 only
  if (tid == (D < THREADS - kWave ? D : 0)) { store_granule(part + D, m_run, tag); store_granule(part + D + 1, l_tot, tag); }
  if (p.poll_merge) {
    if (split != p.S - 1) {
      // CHAIN: the waves that publish nothing would be back at the top of the next layer's step at once, and their tile requests
      // (20 KB per wave) would enter the CU's one vector-memory queue in FRONT of this partial's granules — the merger would see
      // them ~2.5 us late (measured: tools/mb/chain_trace.py).  They wait until the granules have been ISSUED.
      SPATTEN_CSTAMP(8);
      if (CHAIN) __builtin_amdgcn_s_barrier();
      SPATTEN_CSTAMP(9);
      return;
    }
  } else {
    __syncthreads();
    SPATTEN_TSTAMP(3);
    if (s_ticket != (unsigned)(p.S - 1)) return;
  }

  // merge in ONE round trip: thread (g, e) takes splits s = g, g+G, ...; every load below — its partial-o
  // elements and the (m, l) of the same splits — is independent.  Each group folds its splits relative to
  // its own running max; the groups are then folded through LDS.
  constexpr int KB = 8;                          // splits per thread per round trip
  // up to KB splits: ONE thread group folds them all (no LDS fold, no barrier); more: G groups take every G-th split
  const int Gr = p.S > KB ? G : 1;
  const int e = tid % D, g = tid / D;
  float mg = -INFINITY, lg = 0.f, og = 0.f;
  bool expired = false;
  if (g < Gr) {
    for (int s0 = g; s0 < p.S; s0 += KB * Gr) {
      unsigned long long ga[KB], gm[KB], gl[KB];
      int spins = 0;
      bool landed;
      do {   // every load is issued before any tag is looked at: ONE round trip (a short-circuiting `&&` chain makes
             // the compiler wait for each split's granules before it loads the next split's)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const int sc_ = (s0 + k * Gr) < p.S ? (s0 + k * Gr) : g;
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
      } while (!landed && ++spins < (1 << 16));   // bounded: a granule that was issued always lands — if not, fail loudly
      expired |= !landed;
      float a[KB], ms[KB], ls[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const bool live = (s0 + k * Gr) < p.S;
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
  SPATTEN_CSTAMP(6);
  if (expired) {   // never merge incomplete data silently: flag the workspace and poison this unit's output
    atomicOr(p.ws_err, 1u);
    og = __builtin_nanf("");
  }
  if (Gr > 1) {                                  // fold the thread groups through LDS
    if (g < Gr) { s_o[g][e] = og; if (e == 0) { s_o[g][D] = mg; s_o[g][D + 1] = lg; } }
    __syncthreads();
    if (g == 0) {
      float mn = mg;
#pragma unroll
      for (int gg = 1; gg < Gr; ++gg) mn = fmaxf(mn, s_o[gg][D]);
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu);
      og *= w0; lg *= w0;
#pragma unroll
      for (int gg = 1; gg < Gr; ++gg) {
        const float w = __expf(s_o[gg][D] - mu);
        og = fmaf(s_o[gg][e], w, og);
        lg = fmaf(s_o[gg][D + 1], w, lg);
      }
      mg = mn;
    }
  }
  if (!SCORES_ONLY && g == 0) outp[e] = DT<T>::from_f32(og / lg);
  // OPROJ: the merged head goes out to every workgroup's projection team as well (value = what `out` holds)
  if (OPROJ && g == 0) store_granule(p.ych + (int64_t)unit * D + e, DT<T>::round(og / lg), (gen_y & 0x7FFFFFFFu) + 1u);
  if (!SCORES_ONLY && p.head_abs != nullptr) {
    if (KSRC == 1 && tid == 0) s_ticket = (1.0f / lg) < p.pq_thr ? 1u : 0u;               // reuse the ticket word
    if (KSRC == 1) __syncthreads();
    add_head_abs(og / lg, g == 0, !(KSRC == 1 && s_ticket != 0u));
  }
  if (tid == 0) {
    if (CHAIN) chain_complete(p, b);
    if (lse_cur != nullptr) { float* ls = lse_cur + ((int64_t)(b * p.H + h) * p.lse_q + qi) * 2; ls[0] = mg; ls[1] = lg; }
    if (KSRC == 1) p.pq_need[unit] = (1.0f / lg) < p.pq_thr ? 1 : 0;
    p.ws_cnt[2 * unit + 1] = gen + 1u;                                                         // next launch: new tag
    if (!p.poll_merge) __hip_atomic_store(p.ws_cnt + 2 * unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm the counter
  }
  SPATTEN_TSTAMP(4);
  SPATTEN_CSTAMP(7);
  // CHAIN: the other waves of the merging workgroup wait here for the merge: a CU's vector-memory path is one queue, and the
  // next layer's tile requested by them now (20 KB per wave) would sit in front of the merge's polls and of the `out` / completion
  // stores (measured: partials seen 4 us after the last one was published, tools/mb/chain_trace.py)
  if (CHAIN) __builtin_amdgcn_s_barrier();
}

}  // namespace spatten

