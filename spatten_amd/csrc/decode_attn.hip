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
// rounding boundary.  P IS rounded to the model dtype before P·V, like the reference's `.to(query_states.dtype)`
// (:135-138), but UN-NORMALISED: the kernel rounds exp(s - running max) (PairFma::pack_p feeds the packed-dot units),
// the reference rounds exp(s - max) / sum — the global denominator is not known before the first V row.  fp32 keeps
// P in fp32.  Tolerance stated in tests/util.py.
#include "decode_body.h"

namespace spatten {

template <typename T, int D, int UNR, int MODE = 0, bool LEAN = false, int KSRC = 0, bool NT = LEAN, bool CASC = false,
          bool PIPE = false, bool DYN = false, int THREADS = kDecodeThreads>
__global__ __launch_bounds__(THREADS) void decode_attn_kernel(const DecodeParams<T> p) {
  decode_body<T, D, UNR, MODE, LEAN, KSRC, NT, CASC, PIPE, DYN, false, false, THREADS>(p);
}

// The plain decode step with its launch-critical arguments FIRST and 32-bit strides: built with
// -amdgpu-kernarg-preload-count=16 (Makefile) the first 16 kernel-argument dwords — everything the query and the K/V
// tile loads need (q is dense [B,H,D] here) — arrive in SGPRs with the wave, so those loads are issued without waiting
// for a scalar load of the argument block (the other arguments are fetched while they are in flight).
template <typename T, int D, int UNR, bool CASC, bool PIPE, bool DYN = false, int THREADS = kDecodeThreads>
__global__ __launch_bounds__(THREADS) void decode_lean_kernel(T* krc, T* vc, const T* q, const T* cos, const T* sin,
                                                                     int kv_sb, int kv_sh, int N, int chunk, int H, int pos_q,
                                                                     const DecodeParams<T> rest) {
  DecodeParams<T> p = rest;
  p.krc = krc; p.vc = vc; p.q = q; p.cos = cos; p.sin = sin;
  p.kv_sb = kv_sb; p.kv_sh = kv_sh; p.q_sb = (int64_t)H * D; p.q_sh = D;
  p.N = N; p.chunk = chunk; p.H = H; p.pos_q = pos_q;
  decode_body<T, D, UNR, 0, true, 0, true, CASC, PIPE, DYN, false, false, THREADS>(p);
}

// The same with a head list (round 5): blockIdx.y -> head_ids[blockIdx.y].  The pointer rides among the preloaded arguments
// (H and pos_q arrive with the rest of the block), so the id is requested by the wave's first instruction.
template <typename T, int D, int UNR, bool DYN = false, int THREADS = kDecodeThreads>
__global__ __launch_bounds__(THREADS) void decode_lean_hids_kernel(T* krc, T* vc, const T* q, const int32_t* head_ids, const T* cos,
                                                                          const T* sin, int kv_sb, int kv_sh, int N, int chunk, int H,
                                                                          int pos_q, const DecodeParams<T> rest) {
  DecodeParams<T> p = rest;
  p.krc = krc; p.vc = vc; p.q = q; p.cos = cos; p.sin = sin; p.head_ids = head_ids;
  p.kv_sb = kv_sb; p.kv_sh = kv_sh; p.q_sb = (int64_t)H * D; p.q_sh = D;
  p.N = N; p.chunk = chunk; p.H = H; p.pos_q = pos_q;
  decode_body<T, D, UNR, 0, true, 0, true, false, false, DYN, false, false, THREADS, true>(p);
}

// ------------------------------------------------------------------------------------------------
// decode_qkv_kernel — the layer-step's q / k / v projections AND its attention in ONE launch (round 4).
//
// modify_llama.py:72-74 (q_proj / k_proj / v_proj of the single-token row) + :86-147.  As separate launches the step is
// [stacked q/k/v GEMV 17.3 us] -> [attention 11.7 us]: the attention launch is latency-bound (5.4 us of bytes, 3.8 us of
// launch / first-byte latency, 2.4 us of reduction + merge) and its K/V stream does not depend on the query at all.  Here a
// workgroup (split s of head h, 512 threads) is two teams:
//   waves 4-7  project the 3 * D / S elements of q_h, k_h, v_h that fall to this split — rows of the stacked weight, streamed
//              exactly like gemv.hip (lane = 8 columns of a 512-column chunk, 16-byte loads, packed dots, the SAME summation
//              order: bit-identical values), two rows per pass, double buffered — publish them as {value, tag} granules and
//              one wave gathers the head's 3 * D elements from its S splits (one memory hop) into LDS;
//   waves 0-3  are the plain decode step (decode_body<FUSED>): they wait (B1) until the projection waves have ISSUED their
//              last weight pass, issue their whole K/V tile behind it — the memory pipe never drains between the two
//              streams, and the q hand-over happens while the tile is in flight — then (B2) take q / k / v from LDS and run
//              the step's arithmetic, reduction, publication and merge unchanged.
// One kernel boundary and one launch + first-byte latency per layer-step disappear; the K/V bytes travel at the streaming
// rate instead of the latency-bound one.  Requirements (else SPATTEN_ERR_UNSUPPORTED and the caller launches the two
// kernels): the lean step (MHA, one query row, no mask / position tensor / head list / head importance), 16-bit dtype,
// d = 128, a single-shot tile (<= 320 rows per split), S in {1, 2, 4, 8} with the polling merge (the barrier count of the
// two teams must match: B1, B2 and the reduction's one LDS hop).
// ------------------------------------------------------------------------------------------------
template <typename T, int D, bool OPROJ>
__device__ __forceinline__ void qkv_projection_waves(const DecodeParams<T>& p, float* s_x, T* s_y) {
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  using D8 = Dot8<T>;
  constexpr int C = kGemvChunksFused;            // 512-column chunks per pass (gemv.hip: 8)
  const int tid = (int)threadIdx.x - kDecodeThreads;
  const int lane = tid & 63, g = tid >> 6;
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int unit = b * p.H + h;
  const int eps = D / p.S;                       // elements of each of q, k, v projected by this split
  const int rpw = 3 * eps / 4;                   // weight rows per wave (12 at S = 8): a multiple of 2
  const int K = p.hidden, n_kb = (K + C * 512 - 1) / (C * 512);
  const int n_units = (rpw / 2) * n_kb;          // (row pair, column block) passes: even
  unsigned long long* xch = p.xch + (int64_t)unit * (3 * D);
  const unsigned gen = p.ws_cnt[(p.S > 1 ? 2 * unit + 1 : 0) + opaque_lane(0)];
  const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;

  // passes of two weight rows x one 4096-column block, two in flight (r04 A/B: a third pass in flight changes nothing —
  // the projection team alone streams at the stand-alone spatten_gemv rate; what its 18.9 us against 17.3 hold is the
  // exchange hop); x is loaded once when the hidden size is one column block (Llama-2-7B), per pass otherwise
  struct Pass { raw_t w[2][C]; };
  Pass pa, pb;
  raw_t xr[C];
  auto row_of = [&](int rho) {                   // local row -> (which projection, element of the head)
    const int m = rho / eps, e = eps * split + rho % eps;
    return m * D + e;                            // index into the head's q | k | v
  };
  auto load_x = [&](int k0) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      xr[c] = V8::ldg(p.x + b * (int64_t)K + min(k0 + c * 512 + lane * 8, K - 8));
    }
  };
  auto issue = [&](Pass& ps, int u) {
    const int rp = u / n_kb, k0 = (u % n_kb) * (C * 512);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = row_of(rpw * g + 2 * rp + r);
      const T* wrow = p.wqkv + ((int64_t)(idx / D) * p.H * D + (int64_t)h * D + idx % D) * p.w_sn;
#pragma unroll
      for (int c = 0; c < C; ++c) ps.w[r][c] = V8::ldg_stream(wrow + min(k0 + c * 512 + lane * 8, K - 8));
    }
  };
  float acc[2] = {0.f, 0.f};
  auto compute = [&](Pass& ps, int u) {
    const int rp = u / n_kb, kb = u % n_kb, k0 = kb * (C * 512);
    if (n_kb > 1 || K % 8 != 0 || K < C * 512) {
      if (n_kb > 1) load_x(k0);                  // (L2 hits; the wave waits for them here)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (k0 + c * 512 + lane * 8 >= K) {      // columns past K: the x piece is zeroed (gemv.hip)
          float z[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) z[i] = 0.f;
          xr[c] = V8::pack(z);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[r] = D8::dot(xr[c], ps.w[r][c], acc[r]);
    if (kb == n_kb - 1) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int idx = row_of(rpw * g + 2 * rp + r);
        float v = wave_sum(acc[r]);
        if (p.qkv_bias) v += DT<T>::to_f32(p.qkv_bias[(int64_t)(idx / D) * p.H * D + (int64_t)h * D + idx % D]);
        v = DT<T>::round(v);                     // nn.Linear's one rounding to the model dtype
        if (lane == 0) store_granule(xch + idx, v, tag);
        acc[r] = 0.f;
      }
    }
  };
#ifndef SPATTEN_FUSED_B1
#define SPATTEN_FUSED_B1 0
#endif
  // B1 (a bare s_barrier: no vmcnt drain) tells the attention waves to issue their K/V tile.  Where: 0 = when the LAST weight
  // pass has been issued (the tile queues behind two passes in flight), 1 = when only the last pass is still in flight,
  // 2 = when every weight has landed (the memory pipe drains for one round trip, but the tile takes no bandwidth from the
  // weights the query waits for)
  SPATTEN_TSTAMP_T(8, kDecodeThreads);
  if (n_kb == 1) load_x(0);
  issue(pa, 0);
  for (int u = 0; u < n_units; u += 2) {         // n_units is even
    issue(pb, u + 1);
    if (SPATTEN_FUSED_B1 == 0 && u + 2 >= n_units) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(pa, u);
    if (u + 2 < n_units) issue(pa, u + 2);
    if (SPATTEN_FUSED_B1 == 1 && u + 2 >= n_units) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(pb, u + 1);
  }
  if (SPATTEN_FUSED_B1 == 2) __builtin_amdgcn_s_barrier();
  SPATTEN_TSTAMP_T(9, kDecodeThreads);           // every weight consumed, the split's elements published
  if (g == 0) {       // gather the head's q | k | v from its splits: 3 * D granules, 3 * D / 64 per lane, one round trip
    constexpr int PER = 3 * D / kWave;
    unsigned long long gr[PER];
    int spins = 0;
    bool landed;
    do {
      unsigned diff = 0u;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        gr[k] = __hip_atomic_load(xch + lane + kWave * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        diff |= (unsigned)(gr[k] >> 32) ^ tag;
      }
      landed = __all(diff == 0u);
    } while (!landed && ++spins < (1 << 16));
    if (!landed) atomicOr(p.ws_err, 1u);
#pragma unroll
    for (int k = 0; k < PER; ++k) s_x[lane + kWave * k] = landed ? __uint_as_float((unsigned)gr[k]) : __builtin_nanf("");
  }
  SPATTEN_TSTAMP_T(10, kDecodeThreads);          // the head's q | k | v gathered
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS stores above have completed
  __builtin_amdgcn_s_barrier();                  // B2: q | k | v of the head are in LDS
  if (!OPROJ) {
    __builtin_amdgcn_s_barrier();                // B3: the reduction's LDS hop of the attention waves (decode_body)
    return;
  }
  // ---- the output projection of the step (modify_llama.py:163; gemv.hip's mapping: this workgroup's 16 rows, 4 per wave, one
  // pass over the H * D columns).  The weights do not depend on anything: they are requested NOW — the attention team is busy
  // with its tile, reduction and merge for the next ~5 us, during which they stream — and wait in registers for the merged
  // heads of ALL workgroups.
  const int wg = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
  const int n0 = (wg * 4 + g) * 4;               // first of this wave's 4 output rows
  const int Ko = p.H * D;                        // = C * 512 (checked on the host)
  raw_t ow[4][C];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const T* wrow = p.ow + (int64_t)min(n0 + r, p.n_out - 1) * p.ow_sn;
#pragma unroll
    for (int c = 0; c < C; ++c) ow[r][c] = V8::ldg_stream(wrow + c * 512 + lane * 8);
  }
  const unsigned gy = p.ych_gen[opaque_lane(0)];
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();                  // B3: the reduction's LDS hop of the attention waves (decode_body)
  const unsigned tagy = (gy & 0x7FFFFFFFu) + 1u;
  SPATTEN_TSTAMP_T(11, kDecodeThreads);          // B3 passed (the o_proj weights were requested before it)
  if (g == 0) {       // gather the merged attention outputs of ALL heads: H * D granules, 16 per lane and sweep (r04 A/B: a quarter
                      // per wave of the team is SLOWER, 37.5 against 36.4 us per layer — four times the polls in front of the merge)
    bool fail = false;
    for (int base = 0; base < Ko; base += 16 * kWave) {
      unsigned long long gr[16];
      int spins = 0;
      bool landed;
      do {
        unsigned diff = 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          gr[k] = __hip_atomic_load(p.ych + base + lane + kWave * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          diff |= (unsigned)(gr[k] >> 32) ^ tagy;
        }
        landed = __all(diff == 0u);
        if (!landed) __builtin_amdgcn_s_sleep(1);
      } while (!landed && ++spins < (1 << 18));
      fail |= !landed;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        s_y[base + lane + kWave * k] = DT<T>::from_f32(landed ? __uint_as_float((unsigned)gr[k]) : __builtin_nanf(""));
    }
    if (fail) atomicOr(p.ws_err, 1u);
  }
  SPATTEN_TSTAMP_T(12, kDecodeThreads);          // the merged heads of every workgroup are in LDS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // B4 (the attention waves have terminated by now: only this team is counted)
  asm volatile("" ::: "memory");
  {
    raw_t xo[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xo[c] = *reinterpret_cast<const raw_t*>(s_y + c * 512 + lane * 8);
    float oacc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      oacc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) oacc[r] = D8::dot(xo[c], ow[r][c], oacc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) oacc[r] = wave_sum(oacc[r]);
    if (lane < 4 && n0 + lane < p.n_out) {
      float v = oacc[0];
#pragma unroll
      for (int r = 1; r < 4; ++r) v = (lane == r) ? oacc[r] : v;
      if (p.o_bias) v += DT<T>::to_f32(p.o_bias[n0 + lane]);
      p.y[n0 + lane] = DT<T>::from_f32(v);
    }
  }
  SPATTEN_TSTAMP_T(13, kDecodeThreads);          // this workgroup's 16 output rows are stored
  // every workgroup has read this launch's generation before any head could be merged: advance it for the next launch
  if (wg == 0 && tid == 0) p.ych_gen[0] = gy + 1u;
}

template <typename T, int D, int UNR, bool DYN, bool OPROJ>
__global__ __launch_bounds__(2 * kDecodeThreads) void decode_qkv_kernel(T* krc, T* vc, const T* cos, const T* sin, int kv_sb,
                                                                        int kv_sh, int N, int chunk, int H, int pos_q,
                                                                        const DecodeParams<T> rest) {
  __shared__ float s_x[3 * D];
  __shared__ __attribute__((aligned(16))) T s_y[OPROJ ? kGemvChunksFused * 512 : 8];
  DecodeParams<T> p = rest;
  p.krc = krc; p.vc = vc; p.cos = cos; p.sin = sin;
  p.kv_sb = kv_sb; p.kv_sh = kv_sh; p.q_sb = (int64_t)H * D; p.q_sh = D;
  p.N = N; p.chunk = chunk; p.H = H; p.pos_q = pos_q;
  if (threadIdx.x >= kDecodeThreads) {
    qkv_projection_waves<T, D, OPROJ>(p, s_x, s_y);
    return;
  }
#if defined(SPATTEN_FUSED_EXP) && SPATTEN_FUSED_EXP == 1     // A/B harness only: what does the projection team alone take?
  __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
  return;
#endif
  decode_body<T, D, UNR, 0, true, 0, true, false, false, DYN, true, OPROJ>(p, s_x);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// row-groups per tile.  Single-shot instantiation: 16-bit dtypes 10 (d = 128: 320 rows — a whole Llama-2-7B split — in
// flight per workgroup).  Pipelined instantiation (chunks longer than that): two tiles of 4 (fp32: 2) row-groups.
template <typename T, int D> constexpr int decode_unr() { return sizeof(T) == 4 ? (D == 256 ? 2 : 4) : (D == 256 ? 4 : 10); }
#ifndef SPATTEN_DECODE_UP      // row-groups per pipelined tile, 16-bit dtypes.  r03 A/B at 16384 rows x 40 heads (tools/mb/dec_exp.sh
                               // "-DSPATTEN_DECODE_UP=n"): 3: 55.7-56.5 us, 4: 55.3, 5: 56.2-56.7, 6: 58.1-58.7
#define SPATTEN_DECODE_UP 4
#endif
template <typename T, int D> constexpr int decode_unr_pipe() { return sizeof(T) == 4 ? 2 : SPATTEN_DECODE_UP; }
static inline int decode_group_rows(int d) { return kDecodeThreads / (d / 16); }

// threads of the attention team of a single-row, single-shot decode step: 512 (two waves per SIMD, round 4) or 256
static std::atomic<int> g_decode_team{0};
int decode_team() {   // (C++ linkage: decode_chain.hip follows the same team)
  int t = g_decode_team.load(std::memory_order_relaxed);
  if (t == 0) {
    const char* e = getenv("SPATTEN_DECODE_TEAM");
    t = (e && atoi(e) == 256) ? 256 : 512;
    g_decode_team.store(t, std::memory_order_relaxed);
  }
  return t;
}
extern "C" int spatten_decode_set_team(int threads) {
  if (threads != 256 && threads != 512) return SPATTEN_ERR_INVALID;
  const int prev = decode_team();
  g_decode_team.store(threads, std::memory_order_relaxed);
  return prev;
}

int decode_auto_splits(int units, int d, int kv_len, int elt, bool dense_rule) {
  // one workgroup per CU (the device's own count: ADVICE r05): measured best at Llama-2-7B decode sizes — more splits shorten
  // each workgroup's stream but lengthen the merge (a memory round trip per batch of partials)
  int s = coresident_workgroups() / (units > 0 ? units : 1);
  static int env_s = -1;
  if (env_s < 0) { const char* e = getenv("SPATTEN_DECODE_SPLITS"); env_s = e ? atoi(e) : 0; }
  if (env_s > 0) s = env_s;
  const int max_by_len = ceil_div(kv_len, 2 * decode_group_rows(d));   // at least two row-groups per split
  if (s > max_by_len) s = max_by_len;
  // few units (a head-parallel rank holding 4-8 heads): the launch is pure latency, and every 16 further splits are one
  // more round trip of the merge — r03, 4 heads x 2081 rows: 33 splits 9.7 us, 16 splits 8.45; 5 heads x 8192 rows: 51
  // splits 12.4 us, 32 splits 10.9 (tools/probe_few_heads.py).  One round unless the rows are long enough to pay for more.
  const int cap_by_merge = std::max(16, ceil_div(kv_len, 256));
  if (env_s <= 0 && s > cap_by_merge) s = cap_by_merge;
  // r05 (tools/mb/split_sweep.py, 2081 rows): whenever EIGHT splits still hold their chunk in one single-shot tile, eight beat every
  // larger count — the merge folds up to 8 partials in one thread group (no LDS fold, no barrier) and a fuller tile costs nothing
  // while every load is issued up front: 4 / 8 / 16 / 24 / 28 heads 9.30 / 9.56 / 10.09 / 11.29 / 12.11 -> 8.84 / 9.04 / 9.58 /
  // 10.09 / 10.58 us (32 heads were at 8 already).  Longer rows (pipelined tiles) keep one workgroup per CU.
  // (dense_rule: measured on the 16-bit dense step only — the quantised-plane passes have other tile sizes and keep their count)
  if (env_s <= 0 && dense_rule && elt == 2 && d != 256 && s > 8 && ceil_div(kv_len, 8) <= 10 * decode_group_rows(d)) s = 8;
  if (s < 1) s = 1;
  if (s > kDecodeMaxSplits) s = kDecodeMaxSplits;
  return s;
}

// the fused projection + attention launch: the two teams of a workgroup must meet the same number of barriers (S <= 8 with the
// polling merge, or one split), and a split projects a whole number of row pairs per wave
static inline bool decode_qkv_shape_ok(int d, int S, int poll_merge) {
  if (S != 1 && S != 2 && S != 4 && S != 8) return false;
  if (S > 1 && !poll_merge) return false;
  return d % S == 0 && (3 * (d / S)) % 8 == 0;
}

template <typename T, int D>
static int launch_decode(const DecodeParams<T>& p, int n_active, bool scores_only, hipStream_t stream) {
  constexpr int U = decode_unr<T, D>();
  constexpr int UP = decode_unr_pipe<T, D>();
  const dim3 grid((unsigned)p.S, (unsigned)n_active, (unsigned)(p.B * p.n_q));
  const dim3 blk(kDecodeThreads);
  // a chunk that fits one single-shot tile: everything in flight at once; longer: the double-buffered loop
  const bool pipe = p.chunk > U * (kDecodeThreads / (D / 16)) && p.n_q == 1 && !scores_only;
  const bool casc = p.acc != nullptr;
  const bool dyn = p.step != nullptr;      // decode_rows admits it for the plain single-row step only
#define SPATTEN_LAUNCH(...) hipLaunchKernelGGL((decode_attn_kernel<T, D, __VA_ARGS__>), grid, blk, 0, stream, p)
  if (scores_only) {
    // no V tile: the pipelined instantiation affords twice the key rows per tile (first pass of local V pruning at long
    // context, where a split is many tiles long)
    if (p.chunk > U * (kDecodeThreads / (D / 16)) && p.n_q == 1 && sizeof(T) == 2 && D <= 128) SPATTEN_LAUNCH(2 * UP, 1, false, 0, true, false, true);
    else SPATTEN_LAUNCH(U, 1);
  } else if (p.pq_msb != nullptr) {   // progressive-quant keys: pass 1 on the MSB plane, then the refetch pass (same grid)
    if constexpr (D == 256) return SPATTEN_ERR_UNSUPPORTED;
    else {
      // the fused cascade accumulation rides on pass 1 only (pass 2 re-streams the flagged heads: no double count)
      if (dyn) {      // device-length form (the step's row appended by spatten_kv_append_step)
        if (pipe) {
          if (casc) SPATTEN_LAUNCH(UP, 0, false, 1, true, true, true, true); else SPATTEN_LAUNCH(UP, 0, false, 1, true, false, true, true);
          SPATTEN_LAUNCH(UP, 0, false, 2, true, false, true, true);
        } else {
          if (casc) SPATTEN_LAUNCH(U, 0, false, 1, true, true, false, true); else SPATTEN_LAUNCH(U, 0, false, 1, true, false, false, true);
          SPATTEN_LAUNCH(U, 0, false, 2, true, false, false, true);
        }
      } else if (pipe) {
        // (r03, measured and NOT adopted: 6 / 8 row-groups per pipelined tile for this pass — a plane row is a quarter of
        //  a 16-bit key row, so the registers are there — 22.9 / 23.3 us against 20.7 with 4; SPATTEN_PQ_UP keeps the A/B)
        constexpr int UPQ = (sizeof(T) == 2 && D <= 128) ? SPATTEN_PQ_UP : UP;
        if (casc) SPATTEN_LAUNCH(UP, 0, false, 1, true, true, true); else SPATTEN_LAUNCH(UPQ, 0, false, 1, true, false, true);
        SPATTEN_LAUNCH(UP, 0, false, 2, true, false, true);
      } else {
        if (casc) SPATTEN_LAUNCH(U, 0, false, 1, true, true); else SPATTEN_LAUNCH(U, 0, false, 1, true);
        SPATTEN_LAUNCH(U, 0, false, 2, true);
      }
    }
  } else {
    const bool lean = p.n_q == 1 && p.Hkv == p.H && !p.mask && !p.pos_ids && !p.head_ids && !p.causal &&
                      (p.x != nullptr || (p.q_sh == D && p.q_sb == (int64_t)p.H * D));
    const int64_t lim = 0x7FFFFFFF;
    const bool small = p.kv_sb <= lim && p.kv_sh <= lim;
    if (p.x != nullptr) {   // the step's projections fused into the launch (decode_qkv_kernel)
      if constexpr (sizeof(T) == 2 && D == 128) {
        if (!decode_qkv_shape_ok(D, p.S, p.poll_merge) || !lean || !small || pipe || casc || !p.append || p.head_abs || !p.xch)
          return SPATTEN_ERR_UNSUPPORTED;
        const dim3 blk2(2 * kDecodeThreads);
#define SPATTEN_QKV(DY, OP) hipLaunchKernelGGL((decode_qkv_kernel<T, D, U, DY, OP>), grid, blk2, 0, stream, p.krc, p.vc, p.cos, p.sin, \
                                               (int)p.kv_sb, (int)p.kv_sh, p.N, p.chunk, p.H, p.pos_q, p)
        if (p.ow != nullptr) { if (dyn) SPATTEN_QKV(true, true); else SPATTEN_QKV(false, true); }
        else { if (dyn) SPATTEN_QKV(true, false); else SPATTEN_QKV(false, false); }
#undef SPATTEN_QKV
        return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
      } else {
        return SPATTEN_ERR_UNSUPPORTED;
      }
    }
    // TEAM 512 (round 4): the single-shot tile of a single-row step on TWO waves per SIMD — 512 threads, half as many row groups
    // per thread (a wave issues a vector instruction every ~9 cycles however many waves share its SIMD,
    // profiles/r04_prefill_anatomy.txt D; part of the step's ~2,900 vector instructions per wave is issue time): 11.5 -> 11.0 us at
    // the headline shape.  Every route of such a step takes the same team (lean / general, cascade accumulation, device
    // length) — a step must not change its bits with its route; pipelined (long) chunks, the quantised-key passes and the
    // fused projection launch keep the 256-thread body.  spatten_decode_set_team(256) restores the r03 form process-wide.
    if constexpr (sizeof(T) == 2 && D == 128) {
      static int env_pipe512 = -1;      // (A/B) the pipelined form of long chunks on the 512-thread team as well
      if (env_pipe512 < 0) { const char* e = getenv("SPATTEN_DECODE_TEAM_PIPE"); env_pipe512 = e ? atoi(e) : 0; }
      if (decode_team() == 512 && pipe && env_pipe512 && p.n_q == 1) {
        constexpr int UP2 = (UP + 1) / 2;
        const dim3 blk512(2 * kDecodeThreads);
#define SPATTEN_LEAN512P(CC, DD)                                                                                                    \
  hipLaunchKernelGGL((decode_lean_kernel<T, D, UP2, CC, true, DD, 2 * kDecodeThreads>), grid, blk512, 0, stream, p.krc, p.vc, p.q, p.cos, \
                     p.sin, (int)p.kv_sb, (int)p.kv_sh, p.N, p.chunk, p.H, p.pos_q, p)
#define SPATTEN_GEN512P(CC, DD) \
  hipLaunchKernelGGL((decode_attn_kernel<T, D, UP2, 0, false, 0, true, CC, true, DD, 2 * kDecodeThreads>), grid, blk512, 0, stream, p)
        if (lean && small) {
          if (dyn) { if (casc) SPATTEN_LEAN512P(true, true); else SPATTEN_LEAN512P(false, true); }
          else { if (casc) SPATTEN_LEAN512P(true, false); else SPATTEN_LEAN512P(false, false); }
        } else {
          if (dyn) { if (casc) SPATTEN_GEN512P(true, true); else SPATTEN_GEN512P(false, true); }
          else { if (casc) SPATTEN_GEN512P(true, false); else SPATTEN_GEN512P(false, false); }
        }
#undef SPATTEN_LEAN512P
#undef SPATTEN_GEN512P
        return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
      }
      if (decode_team() == 512 && !pipe && p.n_q == 1) {
        constexpr int U2 = (U + 1) / 2;
        const dim3 blk512(2 * kDecodeThreads);
        // a head list (head pruning; a head-parallel rank's surviving heads) on the lean step (r05): the general kernel cost
        // the 24-of-32-heads launch more than the heads it skipped saved
        const bool lean_h = p.Hkv == p.H && !p.mask && !p.pos_ids && p.head_ids && !p.causal && !p.x && !casc &&
                            p.q_sh == D && p.q_sb == (int64_t)p.H * D;
        if (lean_h && small) {
#define SPATTEN_LEANH512(DD)                                                                                                         \
  hipLaunchKernelGGL((decode_lean_hids_kernel<T, D, U2, DD, 2 * kDecodeThreads>), grid, blk512, 0, stream, p.krc, p.vc, p.q, p.head_ids, \
                     p.cos, p.sin, (int)p.kv_sb, (int)p.kv_sh, p.N, p.chunk, p.H, p.pos_q, p)
          if (dyn) SPATTEN_LEANH512(true); else SPATTEN_LEANH512(false);
#undef SPATTEN_LEANH512
          return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
        }
#define SPATTEN_LEAN512(CC, DD)                                                                                                     \
  hipLaunchKernelGGL((decode_lean_kernel<T, D, U2, CC, false, DD, 2 * kDecodeThreads>), grid, blk512, 0, stream, p.krc, p.vc, p.q, p.cos, \
                     p.sin, (int)p.kv_sb, (int)p.kv_sh, p.N, p.chunk, p.H, p.pos_q, p)
#define SPATTEN_GEN512(CC, DD) \
  hipLaunchKernelGGL((decode_attn_kernel<T, D, U2, 0, false, 0, true, CC, false, DD, 2 * kDecodeThreads>), grid, blk512, 0, stream, p)
        if (lean && small) {
          if (dyn) { if (casc) SPATTEN_LEAN512(true, true); else SPATTEN_LEAN512(false, true); }
          else { if (casc) SPATTEN_LEAN512(true, false); else SPATTEN_LEAN512(false, false); }
        } else {
          if (dyn) { if (casc) SPATTEN_GEN512(true, true); else SPATTEN_GEN512(false, true); }
          else { if (casc) SPATTEN_GEN512(true, false); else SPATTEN_GEN512(false, false); }
        }
#undef SPATTEN_LEAN512
#undef SPATTEN_GEN512
        return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
      }
    }
    if (lean && small) {
#define SPATTEN_LEAN(UU, CC, PP, DD)                                                                                    \
  hipLaunchKernelGGL((decode_lean_kernel<T, D, UU, CC, PP, DD>), grid, blk, 0, stream, p.krc, p.vc, p.q, p.cos, p.sin,   \
                     (int)p.kv_sb, (int)p.kv_sh, p.N, p.chunk, p.H, p.pos_q, p)
      if (dyn && casc) { if (pipe) SPATTEN_LEAN(UP, true, true, true); else SPATTEN_LEAN(U, true, false, true); }
      else if (dyn) { if (pipe) SPATTEN_LEAN(UP, false, true, true); else SPATTEN_LEAN(U, false, false, true); }
      else if (pipe) { if (casc) SPATTEN_LEAN(UP, true, true, false); else SPATTEN_LEAN(UP, false, true, false); }
      else { if (casc) SPATTEN_LEAN(U, true, false, false); else SPATTEN_LEAN(U, false, false, false); }
#undef SPATTEN_LEAN
    } else if (p.n_q == 1) {
      if (dyn && casc) { if (pipe) SPATTEN_LAUNCH(UP, 0, false, 0, true, true, true, true); else SPATTEN_LAUNCH(U, 0, false, 0, true, true, false, true); }
      else if (dyn) { if (pipe) SPATTEN_LAUNCH(UP, 0, false, 0, true, false, true, true); else SPATTEN_LAUNCH(U, 0, false, 0, true, false, false, true); }
      else if (pipe) { if (casc) SPATTEN_LAUNCH(UP, 0, false, 0, true, true, true); else SPATTEN_LAUNCH(UP, 0, false, 0, true, false, true); }
      else { if (casc) SPATTEN_LAUNCH(U, 0, false, 0, true, true); else SPATTEN_LAUNCH(U, 0, false, 0, true); }
    } else {
      SPATTEN_LAUNCH(UP);      // the rows leg of prefill: several query rows re-read K/V through L2, short tiles
    }
  }
#undef SPATTEN_LAUNCH
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

// shared by the decode entry points and the small-q / fp32 leg of spatten_attn_prefill
int decode_rows(const DecodeCall& c, hipStream_t stream) {
  const bool scores_only = (c.flags & SPATTEN_DECODE_SCORES_ONLY) != 0;
  const PQKeys* pq = c.pq;
  if ((!c.q && !c.qkv_x) || (!c.kr_cache && !pq) || !c.cos || !c.sin || (!scores_only && (!c.out || !c.v_cache))) return SPATTEN_ERR_INVALID;
  if (c.qkv_x && (!c.qkv_w || !c.qkv_xch || c.qkv_hidden <= 0 || c.qkv_hidden % 8 != 0 || c.qkv_w_sn < c.qkv_hidden || c.qkv_w_sn % 8 != 0 ||
                  c.k_new || c.n_q != 1 || !c.k_cache || pq || scores_only || c.batch != 1))
    return SPATTEN_ERR_INVALID;
  if (scores_only && (!c.scores || !c.lse || c.k_new)) return SPATTEN_ERR_INVALID;
  int n_active = c.head_ids ? c.n_active : c.heads;
  if (n_active <= 0 || n_active > c.heads) return SPATTEN_ERR_INVALID;
  if (c.batch <= 0 || c.heads <= 0 || c.kv_heads <= 0 || c.heads % c.kv_heads != 0 || c.kv_len <= 0 || c.pos_q < 0 || c.n_q <= 0)
    return SPATTEN_ERR_INVALID;
  if ((c.k_new == nullptr) != (c.v_new == nullptr)) return SPATTEN_ERR_INVALID;
  if (c.k_new && c.n_q != 1) return SPATTEN_ERR_INVALID;
  if (pq && (!pq->msb || !pq->lsb || !pq->scale || !pq->need || c.k_new || c.n_q != 1 || scores_only))
    return SPATTEN_ERR_INVALID;
  if (c.acc && (!c.prev_scores || !c.prev_lse || c.mask || c.n_q != 1 || c.prev_len < 0 || (!c.step && c.prev_len > c.kv_len) || scores_only))
    return SPATTEN_ERR_INVALID;
  if (c.table_rows < c.kv_len || (!c.position_ids && c.pos_q + c.n_q > c.table_rows)) return SPATTEN_ERR_INVALID;
  // device-resident step state: the plain single-row step only (kv_len is then the BOUND the grid is laid out for)
  if (c.step && (c.n_q != 1 || c.mask || c.position_ids || scores_only || c.causal)) return SPATTEN_ERR_INVALID;
  // ... with the fused cascade accumulation: `scores` / `lse` and `prev_scores` / `prev_lse` are the two buffers that swap
  // roles every step (same strides, rows up to the bound)
  if (c.step && c.acc && (!c.scores || !c.lse || !c.prev_scores || !c.prev_lse || c.pv_sb != c.sc_sb || c.pv_sh != c.sc_sh))
    return SPATTEN_ERR_INVALID;
  if (c.head_dim != 64 && c.head_dim != 128 && c.head_dim != 256) return SPATTEN_ERR_UNSUPPORTED;
  if (c.dtype != SPATTEN_F32 && c.dtype != SPATTEN_F16 && c.dtype != SPATTEN_BF16) return SPATTEN_ERR_INVALID;
  const int units = c.batch * c.heads * c.n_q;          // workspace is indexed by the FULL head id
  const int ws_splits = c.ws_splits > 0 ? c.ws_splits : kDecodeMaxSplits;
  // the length the splits are laid out for: the step's own, or a common layout length of a whole turn (the device-length
  // form is laid out for its bound, kv_len)
  const int lay = (!c.step && c.layout_len > c.kv_len && c.n_q == 1) ? c.layout_len : c.kv_len;
  int S = c.n_splits > 0 ? c.n_splits : decode_auto_splits(c.batch * n_active * c.n_q, c.head_dim, lay, c.dtype == SPATTEN_F32 ? 4 : 2, c.pq == nullptr);
  if (S > lay) S = lay;
  if (S > kDecodeMaxSplits) S = kDecodeMaxSplits;
  // balanced chunks: ceil(N / S) rows per split (rounded up to the 8 rows of a stash line), whatever N is
  const int chunk = ceil_div(ceil_div(lay, S), 8) * 8;
  S = ceil_div(lay, chunk);
  if (S > 1 && (!c.workspace || (size_t)units > c.ws_units || S > ws_splits)) return SPATTEN_ERR_INVALID;
  const size_t cnt_bytes = decode_cnt_bytes(c.ws_units);
  // single-hop merge (see the kernel): only when the grid cannot outnumber the workgroup slots of the chip.  Causal
  // multi-row launches are excluded (their later splits may be empty and return early... they still publish, but keep
  // the conservative protocol there); SPATTEN_DECODE_POLL=0 forces the ticket protocol (A/B measurements).
  static int env_poll = -1;
  if (env_poll < 0) { const char* e = getenv("SPATTEN_DECODE_POLL"); env_poll = e ? atoi(e) : 1; }
  const int poll_merge = (env_poll != 0 && S > 1 && c.n_q == 1 && (long long)S * n_active * c.batch <= coresident_workgroups()) ? 1 : 0;

  // ---- the output projection of the step (modify_llama.py:163) as a second launch of the same call: one C call per
  // layer-step for the host.  (r03: fusing it INTO the launch — projection waves in the decode workgroups that stream the
  // weight rows while the attention runs and wait only for the merged output — was built and measured: 21.2 us against
  // 18.4 for the two launches; the weight stream does hide under the decode step (15.3 us with the wait removed), but the
  // hand-off "all heads merged -> every CU" costs a poll round trip plus an activation read under load, more than the
  // kernel boundary it replaces.  HISTORY.md, r04 §3.9.)
  const bool want_proj = c.proj_w != nullptr;
  if (want_proj && (!c.proj_out || c.proj_n <= 0 || c.n_q != 1 || scores_only || c.proj_w_sn < (int64_t)c.heads * c.head_dim))
    return SPATTEN_ERR_INVALID;
  // grouped-query step on the matrix cores (decode_gqa.hip, round 6): a kv head's rows streamed once for its whole query group
  {
    const int rc_g = decode_gqa_rows(c, stream);
    if (rc_g == SPATTEN_OK)
      return want_proj ? gemv_rows(c.dtype, c.out, c.out_sb, c.proj_w, c.proj_w_sn, c.proj_bias, c.proj_out, c.proj_out_sb, c.batch,
                                   c.proj_n, c.heads * c.head_dim, stream)
                       : SPATTEN_OK;
    if (rc_g != SPATTEN_ERR_UNSUPPORTED) return rc_g;
  }
  // the output projection INSIDE the fused projection + attention launch (round 4): when that launch runs at all, its grid has
  // one workgroup per 16 output rows (gemv.hip's mapping) and the contraction is one 4096-column pass
  static int env_oproj = -1;
  if (env_oproj < 0) { const char* e = getenv("SPATTEN_FUSED_OPROJ"); env_oproj = e ? atoi(e) : 1; }
  const bool proj_inside = want_proj && c.qkv_x && env_oproj != 0 && c.dtype != SPATTEN_F32 && c.head_dim == 128 && !c.step_oproj_off &&
                           c.heads * c.head_dim == kGemvChunksFused * 512 && c.proj_n == 16 * S * n_active * c.batch &&
                           c.proj_w_sn % 8 == 0 && c.proj_out_sb >= 0;
#define SPATTEN_FILL(T)                                                                                  \
  DecodeParams<T> p;                                                                                     \
  p.q = (const T*)c.q; p.q_sb = c.q_sb; p.q_sh = c.q_sh; p.q_sq = c.q_sq;                                \
  p.kc = (T*)c.k_cache; p.krc = (T*)c.kr_cache; p.vc = (T*)c.v_cache; p.kv_sb = c.kv_sb; p.kv_sh = c.kv_sh; \
  p.append = c.k_new != nullptr || c.qkv_x != nullptr;                                                   \
  p.x = (const T*)c.qkv_x; p.wqkv = (const T*)c.qkv_w; p.w_sn = c.qkv_w_sn; p.qkv_bias = (const T*)c.qkv_bias;  \
  p.xch = (unsigned long long*)c.qkv_xch; p.hidden = c.qkv_hidden;                                       \
  p.ow = nullptr; p.ow_sn = 0; p.o_bias = nullptr; p.y = nullptr; p.ych = nullptr; p.ych_gen = nullptr; p.n_out = 0; \
  if (proj_inside) {                                                                                     \
    p.ow = (const T*)c.proj_w; p.ow_sn = c.proj_w_sn; p.o_bias = (const T*)c.proj_bias; p.y = (T*)c.proj_out; p.n_out = c.proj_n; \
    p.ych_gen = (unsigned*)((char*)c.qkv_xch + (size_t)units * 3 * c.head_dim * sizeof(unsigned long long)); \
    p.ych = (unsigned long long*)((char*)p.ych_gen + 256);                                                \
  }                                                                                                      \
  p.k_new = (const T*)(c.k_new ? c.k_new : (c.q ? c.q : c.cos)); p.v_new = (const T*)(c.k_new ? c.v_new : (c.q ? c.q : c.cos)); \
  p.new_sb = c.k_new ? c.new_sb : c.q_sb; p.new_sh = c.k_new ? c.new_sh : c.q_sh;                        \
  p.cos = (const T*)c.cos; p.sin = (const T*)c.sin; p.table_rows = c.table_rows;                         \
  p.step = (const int32_t*)c.step; p.nr_row = (c.kv_len < c.table_rows ? c.kv_len : c.table_rows) - 1;   \
  if (c.step) {   /* the state's staged rotary rows stand in for the table: row 0 = query, row 1 = appended key */ \
    p.cos = (const T*)((const char*)c.step + kStepHeader);                                               \
    p.sin = p.cos + 2 * (c.head_dim / 2); p.table_rows = 2; p.nr_row = 1;                                \
  }                                                                                                      \
  p.pos_ids = c.position_ids; p.pos_sb = c.pos_sb;                                                       \
  p.mask = (const T*)c.mask; p.mask_sb = c.mask_sb; p.mask_sq = c.mask_sq;                               \
  p.out = (T*)c.out; p.out_sb = c.out_sb; p.out_sq = c.out_sq;                                           \
  p.scores = (T*)c.scores; p.sc_sb = c.sc_sb; p.sc_sh = c.sc_sh; p.sc_sq = c.sc_sq;                      \
  p.lse = c.lse; p.lse_q = c.lse_q > 0 ? c.lse_q : c.n_q; p.head_ids = c.head_ids;                        \
  p.pq_msb = pq ? pq->msb : nullptr; p.pq_lsb = pq ? pq->lsb : nullptr; p.pq_scale = pq ? pq->scale : nullptr; \
  p.pl_sb = pq ? pq->pl_sb : 0; p.pl_sh = pq ? pq->pl_sh : 0; p.ps_sb = pq ? pq->sc_sb : 0; p.ps_sh = pq ? pq->sc_sh : 0; \
  p.pq_thr = pq ? pq->threshold : 0.f; p.pq_need = pq ? pq->need : nullptr;                              \
  p.prev_scores = (const T*)c.prev_scores; p.pv_sb = c.pv_sb; p.pv_sh = c.pv_sh; p.prev_lse = c.prev_lse; \
  p.acc = (c.prev_len > 0 || c.step) ? c.acc : nullptr; p.acc_sh = c.acc_sh; p.prev_len = c.prev_len;   \
  p.head_abs = c.head_abs;                                                                               \
  p.ws_err = (unsigned*)c.workspace;                                                                     \
  p.ws_cnt = c.workspace ? (unsigned*)((char*)c.workspace + kDecodeWsHeader) : (unsigned*)c.cos;         \
  p.ws_part = c.workspace ? (unsigned long long*)((char*)c.workspace + kDecodeWsHeader + cnt_bytes) : nullptr; \
  p.ws_unit = (int64_t)ws_splits * (c.head_dim + 2);                                                     \
  p.B = c.batch; p.H = c.heads; p.Hkv = c.kv_heads; p.N = c.kv_len; p.pos_q = c.step ? 0 : c.pos_q; p.S = S; p.chunk = chunk; \
  p.n_q = c.n_q; p.causal = c.causal; p.vis0 = c.vis0 > 0 ? c.vis0 : c.kv_len - c.n_q + 1;                \
  p.poll_merge = poll_merge;                                                                             \
  p.sqrt_d = sqrtf((float)c.head_dim);                                                                   \
  {                                                                                                      \
    const int rc_ = dispatch_decode<T>(p, c.head_dim, n_active, scores_only, stream);                    \
    if (rc_ != SPATTEN_OK || !want_proj || proj_inside) return rc_;                                      \
    return gemv_rows(c.dtype, c.out, c.out_sb, c.proj_w, c.proj_w_sn, c.proj_bias, c.proj_out, c.proj_out_sb, c.batch, \
                     c.proj_n, c.heads * c.head_dim, stream);                                            \
  }

  switch (c.dtype) {
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
  return kDecodeWsHeader + decode_cnt_bytes(units) + units * max_splits * (head_dim + 2) * sizeof(unsigned long long);
}

extern "C" int spatten_decode_auto_splits(int batch, int heads, int head_dim, int kv_len) {
  if (batch <= 0 || heads <= 0 || kv_len <= 0 || (head_dim != 64 && head_dim != 128 && head_dim != 256)) return 1;
  return decode_auto_splits(batch * heads, head_dim, kv_len, 2, true);
}

extern "C" size_t spatten_decode_qkv_exchange_bytes(int batch, int heads, int head_dim) {
  if (batch <= 0 || heads <= 0 || head_dim <= 0) return 0;
  // [units][3 d] granules: a head's q | k | v between its splits; then a 256-byte header (word 0: the generation of the
  // output exchange) and [units][d] granules: the merged heads to every workgroup (the output projection inside the launch)
  return (size_t)batch * heads * 3 * head_dim * sizeof(unsigned long long) + 256 + (size_t)batch * heads * head_dim * sizeof(unsigned long long);
}

extern "C" int spatten_decode_qkv_supported(int dtype, int batch, int heads, int kv_heads, int head_dim, int kv_len_layout) {
  if (dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return 0;
  if (batch != 1 || heads <= 0 || heads != kv_heads || head_dim != 128 || kv_len_layout <= 0) return 0;
  int S = decode_auto_splits(batch * heads, head_dim, kv_len_layout, 2, true);
  if (S > kv_len_layout) S = kv_len_layout;
  const int chunk = ceil_div(ceil_div(kv_len_layout, S), 8) * 8;
  S = ceil_div(kv_len_layout, chunk);
  static int env_poll = -1;
  if (env_poll < 0) { const char* e = getenv("SPATTEN_DECODE_POLL"); env_poll = e ? atoi(e) : 1; }
  const int poll = (env_poll != 0 && S > 1 && (long long)S * heads * batch <= coresident_workgroups()) ? 1 : 0;
  if (!decode_qkv_shape_ok(head_dim, S, poll)) return 0;
  return chunk <= 10 * decode_group_rows(head_dim) ? 1 : 0;      // a single-shot tile per split
}

extern "C" int spatten_decode_workspace_status(void* workspace, void* stream) {
  if (!workspace) return SPATTEN_ERR_INVALID;
  unsigned flag = 0;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(&flag, workspace, sizeof(flag), hipMemcpyDeviceToHost, st) != hipSuccess) return SPATTEN_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return SPATTEN_ERR_LAUNCH;
  if (flag == 0) return SPATTEN_OK;
  (void)hipMemsetAsync(workspace, 0, sizeof(flag), st);
  return SPATTEN_ERR_TIMEOUT;
}

extern "C" int spatten_attn_decode_args(const spatten_decode_args_t* a, void* stream) {
  if (!a || a->struct_size != sizeof(spatten_decode_args_t)) return SPATTEN_ERR_INVALID;
  DecodeCall c;
  c.dtype = a->dtype;
  c.q = a->q; c.q_sb = a->q_sb; c.q_sh = a->q_sh;
  c.k_cache = a->k_cache; c.kr_cache = a->kr_cache; c.v_cache = a->v_cache; c.kv_sb = a->kv_sb; c.kv_sh = a->kv_sh;
  c.k_new = a->k_new; c.v_new = a->v_new; c.new_sb = a->new_sb; c.new_sh = a->new_sh;
  c.cos = a->cos; c.sin = a->sin; c.table_rows = a->table_rows;
  c.position_ids = a->position_ids; c.pos_sb = a->pos_sb;
  c.mask = a->mask; c.mask_sb = a->mask_sb;
  c.out = a->out; c.out_sb = a->out_sb;
  c.scores = a->scores; c.sc_sb = a->sc_sb; c.sc_sh = a->sc_sh;
  c.lse = a->lse;
  c.workspace = a->workspace; c.ws_units = (size_t)(a->batch > 0 && a->heads > 0 ? (size_t)a->batch * a->heads : 0);
  c.ws_splits = a->workspace_splits > 0 ? a->workspace_splits : kDecodeMaxSplits;
  c.batch = a->batch; c.heads = a->heads; c.kv_heads = a->kv_heads; c.head_dim = a->head_dim; c.kv_len = a->kv_len;
  c.pos_q = a->pos_q; c.n_q = 1; c.causal = 0; c.n_splits = a->n_splits;
  c.head_ids = a->head_ids; c.n_active = a->n_active_heads; c.flags = a->flags;
  c.prev_scores = a->prev_scores; c.pv_sb = a->prev_sb; c.pv_sh = a->prev_sh; c.prev_lse = a->prev_lse;
  c.acc = a->importance_acc; c.acc_sh = a->acc_sh; c.prev_len = a->prev_len;
  c.head_abs = a->head_abs_acc;
  c.step = a->step_state; c.layout_len = a->kv_len_layout;
  c.proj_w = a->proj_weight; c.proj_w_sn = a->proj_w_sn; c.proj_bias = a->proj_bias; c.proj_out = a->proj_out;
  c.proj_out_sb = a->proj_out_sb; c.proj_n = a->proj_n;
  c.qkv_x = a->qkv_x; c.qkv_w = a->qkv_weight; c.qkv_w_sn = a->qkv_w_sn; c.qkv_bias = a->qkv_bias; c.qkv_xch = a->qkv_exchange;
  c.qkv_hidden = a->qkv_hidden;
  PQKeys keys;
  if (a->pq_msb) {
    if (!a->pq_lsb || !a->pq_scale || !a->pq_need_lsb) return SPATTEN_ERR_INVALID;
    if (a->head_dim != 64 && a->head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
    keys.msb = (const uint8_t*)a->pq_msb; keys.lsb = (const uint8_t*)a->pq_lsb; keys.scale = a->pq_scale;
    keys.pl_sb = a->pq_pl_sb; keys.pl_sh = a->pq_pl_sh; keys.sc_sb = a->pq_sc_sb; keys.sc_sh = a->pq_sc_sh;
    keys.threshold = a->pq_threshold; keys.need = a->pq_need_lsb;
    c.pq = &keys;
  }
  return decode_rows(c, (hipStream_t)stream);
}

extern "C" int spatten_attn_decode(int dtype, const void* q, int64_t q_sb, int64_t q_sh, void* k_cache,
                                   void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* k_new,
                                   const void* v_new, int64_t new_sb, int64_t new_sh, const void* cos,
                                   const void* sin, int table_rows, const int64_t* position_ids,
                                   int64_t pos_sb, const void* mask, int64_t mask_sb, void* out,
                                   int64_t out_sb, void* scores, int64_t sc_sb, int64_t sc_sh, float* lse,
                                   void* workspace, int batch, int heads, int kv_heads, int head_dim,
                                   int kv_len, int pos_q, int n_splits, void* stream) {
  spatten_decode_args_t a = {};
  a.struct_size = sizeof(a);
  a.dtype = dtype;
  a.q = q; a.q_sb = q_sb; a.q_sh = q_sh;
  a.k_cache = k_cache; a.kr_cache = kr_cache; a.v_cache = v_cache; a.kv_sb = kv_sb; a.kv_sh = kv_sh;
  a.k_new = k_new; a.v_new = v_new; a.new_sb = new_sb; a.new_sh = new_sh;
  a.cos = cos; a.sin = sin; a.table_rows = table_rows;
  a.position_ids = position_ids; a.pos_sb = pos_sb;
  a.mask = mask; a.mask_sb = mask_sb;
  a.out = out; a.out_sb = out_sb;
  a.scores = scores; a.sc_sb = sc_sb; a.sc_sh = sc_sh;
  a.lse = lse;
  a.workspace = workspace; a.workspace_splits = SPATTEN_DECODE_MAX_SPLITS;
  a.batch = batch; a.heads = heads; a.kv_heads = kv_heads; a.head_dim = head_dim; a.kv_len = kv_len;
  a.pos_q = pos_q; a.n_splits = n_splits;
  return spatten_attn_decode_args(&a, stream);
}

#ifdef SPATTEN_TRACE
extern "C" int spatten_debug_set_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_spatten_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
