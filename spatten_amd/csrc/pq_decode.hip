// pq_decode.hip — the decode step over PROFILED quantised planes (ABI 4): key MSB plane of 4 / 6 / 8 bits + 4-bit LSB plane,
// value plane of 8 / 6 bits (include/spatten.h "Bit profiles and the quantised VALUE plane").  PARITY UNPINNED: the
// reference's Python has no numeric implementation; restated from the RTL and checked against oracle/spatten_oracle.py.
//
//   MatrixFetcher.scala:48-51        supported (bit_count, fused_mat) profiles (4,1) (6,2) (8,1)
//   TestSpAtten.scala:64,83-97,173-176   the harness fetches K at quant_key_bit and V at quant_value_bit (8 by default, 6 in the
//                                    per8 trace); SpAttenController.scala:716-723: V is ONE fetch (`high_bits := True`)
//   SpAttenController.scala:35-39,230-232,296-305   the refetch adds 4 LSBs from a separate plane (write mask 0x00F)
//   RequantDecision.scala:44-72      need_requant = max_j prob_j < threshold;  SpAttenController.scala:402: recompute ONCE
//
// Two launches per step (same grid): PASS 1 scores every key from the MSB plane (logit = 16 (q . msb) scale / sqrt(d), fp32),
// leaves the fp32 MSB logits in planes.msb_logit, runs softmax + P.V over the quantised V and records need_lsb per head in the
// merge step; PASS 2 does work only for the flagged heads: it reads the stashed MSB logit + the LSB plane — logit8 = logit_msb
// + (q . lsb) scale / sqrt(d), the MSB plane is NOT re-read — and streams V again (the probabilities changed).
// HBM bytes per key row, d = 128: profile (4, 8): 64 + 128 + 8 (+ 4 written) in pass 1 and 4 + 64 + 128 + 8 in pass 2,
// against 64 + 256 + 4 and 128 + 256 + 4 with bf16 V (decode_attn.hip KSRC 1 / 2); (8, 8): 128 + 128 + 8; (6, 6): 96 + 96 + 8.
//
// Structure = the pipelined instantiation of decode_attn.hip's decode_body (split-N over one workgroup per CU, per-thread
// online softmax, tiles of UP row-groups double buffered, every load of a tile unconditional and issued before the previous
// tile is processed, the same publish / poll merge of the split partials on the SAME workspace protocol); the launch appends
// nothing (spatten_kv_append[_step] + spatten_pq_pack_planes first).  DYN = the device-resident length (step.hip).
// Every plane is in PIECE order (lane c of a row's D/16 lanes owns 16 elements = its piece): a lane's key piece, value piece
// and query elements line up, so there is no cross-lane traffic before the D/16-lane reduction of the logit.  Loads are
// 32-bit-offset buffer loads; the loop carries no branches per row-group.  r04 first cut (natural-layout nibble plane,
// flat loads with clamped rows, per-row-group branches, IEEE divide per logit): 3,000 instructions per 16 row-groups and
// 19.1 us per launch at 8192 rows x 32 heads — issue-bound, not byte-bound (2.8 TB/s).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "pq_pack.h"

namespace spatten {

template <typename T>
struct PqvParams {
  const T* q; int64_t q_sb, q_sh;
  PlanesDev pl;
  const T* cos; const T* sin; int table_rows;
  T* out; int64_t out_sb;
  T* scores; int64_t sc_sb, sc_sh;
  float* lse;
  const int32_t* head_ids;
  float thr; int32_t* need; float* head_abs; const int32_t* step;
  unsigned long long* ws_part; unsigned* ws_cnt; unsigned* ws_err; int64_t ws_unit;
  int B, H, Hkv, N, pos_q, S, chunk, poll_merge;
  float sqrt_d;
  // the step's APPEND inside the MSB pass (round 5, optional): row N - 1 of kc (optional) / krc / vc <- k_new / v_new [B,Hkv,d]
  // (modify_llama.py:95-104; the key rotated with rotary row nr_row of cos / sin) and that row of every plane
  const T* k_new; const T* v_new; int64_t new_sb, new_sh; T* kc; T* krc; T* vc; int64_t kv_sb, kv_sh; int nr_row;
  // the refetch pass over the FLAGGED heads only (round 6): blockIdx.y = rank among the flagged heads of the launch's n_act heads
  // (every workgroup derives the list itself from `need`: one wave load + ballot, no counter to reset), S / chunk of this pass are
  // its own — many short splits, so that two or three flagged heads are spread over the chip instead of over 8 CUs each
  int compact, n_act, S1, chunk1;                // (S1 / chunk1: pass 1's layout, taken when most heads are flagged)
};

#ifndef SPATTEN_PQV_UP
#define SPATTEN_PQV_UP 4          // row-groups per pipelined tile.  r04 A/B at 8192 rows x 32 heads, profile (4, 8), MSB pass:
                                  // 12: 20.0 us, 8: 16.6, 6: 16.4, 4: 14.8 (tools/mb/pqv_ab.sh; rebuild with -DSPATTEN_PQV_UP=n)
#endif
#ifndef SPATTEN_PQV_THREADS     // A/B: 512 = two waves per SIMD (decode_attn.hip's 512-thread team).  r04 at 8192 rows x 32 heads, MSB pass of
                                // the (4,8) / (8,8) / (6,6) profiles: 256 threads 14.9 / 17.3 / 16.0 us; 512 with UP = 4: 16.9 / 18.2 / 18.3; 512 with
                                // UP = 2: 15.3 / 16.8 / 16.8 — the pipelined stream does not gain from the second wave.  256 stays.
#define SPATTEN_PQV_THREADS 256
#endif
// (r05, measured and dropped: THREE tiles in flight instead of two — (4,8) 14.5 -> 15.3-15.5 us, (8,8) 16.9 -> 18.0-18.2, with tiles of
//  64 rows 14.7 / 17.7-18.1: the pass is not waiting for bytes in flight)
constexpr int kPqvThreads = SPATTEN_PQV_THREADS;
constexpr int kPqvWaves = kPqvThreads / 64;

__device__ inline void pqv_store_granule(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// 32-bit-offset buffer loads (wave-uniform descriptor, per-lane voffset constant over the launch, the tile's row offset in the
// scalar operand): no 64-bit address arithmetic and no row clamping in the loop — a row past `num_records` reads as 0 without
// touching memory.  aux = 2: non-temporal (every plane byte is used once per launch).
template <int W> __device__ inline void bload(__amdgpu_buffer_rsrc_t r, int voff, int soff, uint32_t (&w)[W]) {
  if constexpr (W == 4) {
    const auto x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
  } else if constexpr (W == 3) {
    const auto x = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 2);
    w[0] = x[0]; w[1] = x[1]; w[2] = x[2];
  } else if constexpr (W == 2) {
    const auto x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2);
    w[0] = x[0]; w[1] = x[1];
  } else {
    w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2);
  }
}
__device__ inline float bload_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int W> __device__ inline void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, const uint32_t (&w)[W]) {
  if constexpr (W == 4) {
    const u32x4 x = {w[0], w[1], w[2], w[3]};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, 0);
  } else if constexpr (W == 2) {
    const u32x2 x = {w[0], w[1]};
    __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(w[0], r, voff, soff, 0);
  }
}
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7FFFFFFF ? (bytes > 0 ? bytes : 0) : 0x7FFFFFFF),
                                           0x00020000);
}
// field t (compile-time after unrolling) of a piece of 16 BITS-bit fields held in BITS/2 dwords, as fp32
template <int BITS, int W> __device__ inline float field_f32(const uint32_t (&w)[W], int t) {
  const int bit = BITS * t, wi = bit / 32, off = bit % 32;
  const uint32_t mask = (1u << BITS) - 1u;
  uint32_t f;
  if (off + BITS <= 32) f = (w[wi] >> off) & mask;
  else f = __builtin_amdgcn_alignbit(w[wi + 1 < W ? wi + 1 : wi], w[wi], off) & mask;
  return (float)f;
}
#ifndef SPATTEN_PQV_MAGIC6      // A/B: 6-bit fields through the mantissa (below) instead of extract + convert.  1 = on
#define SPATTEN_PQV_MAGIC6 1
#endif
// 6-bit fields without a conversion (round 5): the 32-bit window that starts at bit `s` of the piece (one shift / v_alignbit) puts
// a field into mantissa bits [P, P + 6) of an fp32 whose exponent makes the mantissa's bit P worth 1 — `(win & mask) | magic`
// (one v_and_or_b32) IS the float 2^(23-P) + field, exactly; the constant leaves through the weight sum like the offset-binary
// bias.  P = 17 (64 + f) and P = 11 (4096 + f) hold two NEIGHBOURING fields of one window: 1.5 instructions per field (values:
// the 4096 costs 6 bits of the accumulator — 2^-12 of a field step per addition, the output tolerance is 2^-7 of full scale);
// P = 17 only: 2 per field (keys: the fp32 logits keep their precision).  r04: shift + and + v_cvt_f32_u32, 3 for a field that
// straddles two dwords = 2.5 per field.
constexpr float kMagic6Hi = 64.0f, kMagic6Lo = 4096.0f;
template <int W> __device__ inline uint32_t piece_window(const uint32_t (&w)[W], int s) {   // bits [s, s + 32) of the piece (s compile-time)
  if (s < 0) return w[0] << (-s);
  const int wi = s / 32, sh = s % 32;
  if (sh == 0) return w[wi];
  if (wi + 1 < W) return __builtin_amdgcn_alignbit(w[wi + 1], w[wi], sh);
  return w[wi] >> sh;
}
__device__ inline float magic6_hi(uint32_t win) { return __uint_as_float(and_or_b32(win, 0x007E0000u, 0x42800000u)); }   // 64 + field at [17, 23)
__device__ inline float magic6_lo(uint32_t win) { return __uint_as_float(and_or_b32(win, 0x0001F800u, 0x45800000u)); }   // 4096 + field at [11, 17)

template <int BITS, int W> __device__ inline float dot_piece(const uint32_t (&w)[W], const float (&qv)[16]) {
  float a0 = 0.f, a1 = 0.f;
  if constexpr (BITS == 6 && SPATTEN_PQV_MAGIC6) {       // every field at [17, 23): sum_t q_t (64 + f_t); the caller takes 64 qsum out
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      a0 = fmaf(qv[t], magic6_hi(piece_window<W>(w, 6 * t - 17)), a0);
      a1 = fmaf(qv[t + 1], magic6_hi(piece_window<W>(w, 6 * (t + 1) - 17)), a1);
    }
    return a0 + a1;
  }
#pragma unroll
  for (int t = 0; t < 16; t += 2) {
    a0 = fmaf(qv[t], field_f32<BITS, W>(w, t), a0);
    a1 = fmaf(qv[t + 1], field_f32<BITS, W>(w, t + 1), a1);
  }
  return a0 + a1;
}
// (magic form: o[t] gains wgt * (4096 + f) for even t, wgt * (64 + f) for odd t: piece_bias6(t) * the weight sum is taken out at the end)
__device__ inline float piece_bias6(int t) { return (t & 1) ? kMagic6Hi : kMagic6Lo; }
template <int BITS, int W> __device__ inline void fma_piece(float (&o)[16], const uint32_t (&w)[W], float wgt) {
  if constexpr (BITS == 6 && SPATTEN_PQV_MAGIC6) {
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const uint32_t win = piece_window<W>(w, 6 * t - 11);     // field t at [11, 17), field t + 1 at [17, 23)
      o[t] = fmaf(wgt, magic6_lo(win), o[t]);
      o[t + 1] = fmaf(wgt, magic6_hi(win), o[t + 1]);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) o[t] = fmaf(wgt, field_f32<BITS, W>(w, t), o[t]);
}

// APP (round 5): the MSB pass that also appends the step's row — its own instantiation: carried as a run-time branch the path cost every
// OTHER caller 0.8-1.0 us per pass (a branch around the new token's loads behind the first tile breaks the exact vmcnt the
// pipelined tiles rely on: same-call A/B, (4,8) 15.5 vs 14.65 us).
template <typename T, int D, int KB, int VB, int PASS, bool DYN, int UP, bool APP = false>
__global__ __launch_bounds__(kPqvThreads) void pqv_decode_kernel(const PqvParams<T> p_in) {
  PqvParams<T> p = p_in;                         // (the compact refetch pass picks its split layout by the number of flagged heads)
  constexpr int LPR = D / 16;                    // lanes per row: lane c owns the row's piece c (16 elements)
  constexpr int RPI = kPqvThreads / LPR;         // rows per row-group
  constexpr int TILE = RPI * UP;
  constexpr int HALF = D / 2;
  constexpr int G = (kPqvThreads / D) > 0 ? (kPqvThreads / D) : 1;
  constexpr int KBITS = (PASS == 1) ? KB : 4;    // pass 2 reads the LSB nibbles
  constexpr int KW = KBITS / 2;                  // dwords of this lane's key piece
  constexpr int KROW = D * KBITS / 8;            // bytes of a key-plane row
  constexpr int VW = VB / 2;
  constexpr int VROW = D * VB / 8;
  using V8 = Vec8<T>;

  __shared__ float s_o[kPqvWaves][D + 2];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x;
  const int c = tid % LPR;
  const int r = tid / LPR;
  const int wave = tid / kWave;
  const int lane = tid % kWave;

  int split = blockIdx.x;
  const int b = blockIdx.z;
  int y = blockIdx.y;
  if (PASS == 2 && p.compact) {
    int f = 0;
    if (lane < p.n_act) f = p.need[b * p.H + (p.head_ids ? p.head_ids[lane] : lane)] != 0;
    unsigned long long m = __ballot(f);
    const int cnt = __popcll(m);
    int rank = blockIdx.y;
    if (2 * cnt > p.n_act) {
      // MANY flagged heads (the every-head-flagged worst case): the fine layout would be 4x the workgroups and 32 partials per
      // merge — the first S1 x cnt workgroups of the grid take pass 1's coarser layout instead, the rest leave
      const int lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
      if (lin >= p.S1 * cnt) return;
      rank = lin / p.S1; split = lin - rank * p.S1;
      p.S = p.S1; p.chunk = p.chunk1;
    } else if (rank >= cnt) {
      return;                                        // fewer flagged heads than this row of the grid
    }
    for (int i = 0; i < rank; ++i) m &= m - 1ull;
    y = __builtin_ctzll(m);
  }
  const int h = p.head_ids ? p.head_ids[y] : y;
  const int hkv = (p.Hkv == p.H) ? h : h / (p.H / p.Hkv);
  const int unit = b * p.H + h;
  if (PASS == 2 && p.need[unit] == 0) return;    // confident head: pass 1 already produced its output

  int n_dyn = 0;
  if (DYN) n_dyn = p.step[opaque_lane(0)];       // requested FIRST (in-order returns), read after the tile loads went out
  const int lo = split * p.chunk;
  const int rl = min(lo + p.chunk, p.N);         // static row limit of the loads (DYN: p.N is the bound): rows >= rl read as 0

  const uint8_t* kpb = (PASS == 1) ? p.pl.km + b * p.pl.km_sb + hkv * p.pl.km_sh : p.pl.kl + b * p.pl.kl_sb + hkv * p.pl.kl_sh;
  float* lgb = p.pl.lg + b * p.pl.lg_sb + h * p.pl.lg_sh;
  const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kpb, (int64_t)rl * KROW);
  const __amdgpu_buffer_rsrc_t rs_ks = make_rsrc(p.pl.ks + b * p.pl.sc_sb + hkv * p.pl.sc_sh, (int64_t)rl * 4);
  const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(p.pl.vq + b * p.pl.vq_sb + hkv * p.pl.vq_sh, (int64_t)rl * VROW);
  const __amdgpu_buffer_rsrc_t rs_vs = make_rsrc(p.pl.vs + b * p.pl.sc_sb + hkv * p.pl.sc_sh, (int64_t)rl * 4);
  const __amdgpu_buffer_rsrc_t rs_lg = make_rsrc(lgb, (int64_t)rl * 4);
  // Rows of a tile (round 5): thread-row r owns the UP CONSECUTIVE rows t0 + r UP + u (r04: t0 + u RPI + r) — a wave's rows of a
  // tile are then one contiguous run, the UP row scales of a lane are ONE 16-byte load per plane (r04: a 4-byte load per
  // row-group and plane — half of the kernel's memory instructions carried 32 bytes per wave), and a lane's UP stash entries /
  // MSB logits leave as one store.  The plane rows of one wave instruction are UP rows apart: still whole 64- / 96- / 128-byte rows.
  static_assert(UP == 4 || UP == 2, "row scales travel as one 8- or 16-byte load");
  const int vo_k = r * (UP * KROW) + c * (2 * KBITS), vo_v = r * (UP * VROW) + c * (2 * VB), vo_s = r * (UP * 4);

  struct Tile {
    uint32_t kw[UP][KW]; uint32_t ks[UP];        // this lane's piece of the key row + the rows' scales (fp32 bits)
    uint32_t vw[UP][VW]; uint32_t vs[UP];
    uint32_t lg[UP];                             // PASS 2: the MSB logits pass 1 left
  };
  Tile tile_a, tile_b;
  auto issue = [&](Tile& tl, int t0) {
    bload<UP>(rs_ks, vo_s, t0 * 4, tl.ks);
    if (PASS == 2) bload<UP>(rs_lg, vo_s, t0 * 4, tl.lg);
#pragma unroll
    for (int u = 0; u < UP; ++u) bload<KW>(rs_k, vo_k, (t0 + u) * KROW, tl.kw[u]);   // (wave-uniform part in the scalar offset)
    bload<UP>(rs_vs, vo_s, t0 * 4, tl.vs);
#pragma unroll
    for (int u = 0; u < UP; ++u) bload<VW>(rs_v, vo_v, (t0 + u) * VROW, tl.vw[u]);
  };

  typename V8::raw q_raw[4];
  {
    const T* qp = p.q + b * p.q_sb + h * p.q_sh;
    const int pq = min(max(p.pos_q, 0), p.table_rows - 1);
    q_raw[0] = V8::ldg(qp + 8 * c);
    q_raw[1] = V8::ldg(qp + HALF + 8 * c);
    q_raw[2] = V8::ldg(p.cos + (int64_t)pq * HALF + 8 * c);
    q_raw[3] = V8::ldg(p.sin + (int64_t)pq * HALF + 8 * c);
  }
  issue(tile_a, lo);
  // APPEND (round 5, MSB pass): every split's first thread-row requests the new token's K / V pieces and the rotary row of its slot
  // behind the first tile (static addresses); the split that owns row N - 1 rotates, stores the cache rows, packs and stores the row
  // of every plane and scores it FROM ITS REGISTERS as one extra key (the tiles never read it; the refetch pass is another launch)
  constexpr bool app = APP && PASS == 1;
  typename V8::raw nk_raw[2], nv_raw[2], nr_raw[2];
  if constexpr (app) {             // (EVERY thread, unconditionally: no control flow between the tile's loads and their first wait)
    const T* kp = p.k_new + b * p.new_sb + hkv * p.new_sh;
    const T* vp = p.v_new + b * p.new_sb + hkv * p.new_sh;
    nk_raw[0] = V8::ldg(kp + 8 * c); nk_raw[1] = V8::ldg(kp + HALF + 8 * c);
    nv_raw[0] = V8::ldg(vp + 8 * c); nv_raw[1] = V8::ldg(vp + HALF + 8 * c);
    nr_raw[0] = V8::ldg(p.cos + (int64_t)p.nr_row * HALF + 8 * c);
    nr_raw[1] = V8::ldg(p.sin + (int64_t)p.nr_row * HALF + 8 * c);
  }
  const unsigned gen = p.ws_cnt[(p.S > 1 ? 2 * unit + 1 : 0) + opaque_lane(0)];
  __builtin_amdgcn_sched_barrier(0);

  const int N = DYN ? __builtin_amdgcn_readfirstlane(n_dyn) : p.N;
  const int hi_all = min(lo + p.chunk, N);
  // (DYN: p.N is the launch's BOUND = the planes' capacity — a replay past it must not write beyond the planes: ADVICE r05)
  const bool owns_new = app && lo <= N - 1 && N - 1 < hi_all && (!DYN || N <= p.N);      // (wave-uniform)
  const int hi = owns_new ? hi_all - 1 : hi_all;                   // rows the tiles score

  // ---- the rotated query: fp32 values (they ARE model-dtype values: rope_pair rounds) in piece order; for the nibble planes
  // also packed for the pair-dot trick (common.h NibbleDot: nibble k of a dword pairs with nibble k + 4)
  float qv[16];
  float qsum = 0.f;
  typename NibbleDot<T>::packed qn[2];
  {
    float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
    V8::unpack(q_raw[0], xlo);
    V8::unpack(q_raw[1], xhi);
    V8::unpack(q_raw[2], cc);
    V8::unpack(q_raw[3], ss);
    rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
#pragma unroll
    for (int i = 0; i < 8; ++i) { qv[i] = ylo[i]; qv[8 + i] = yhi[i]; qsum += ylo[i] + yhi[i]; }
    if (KBITS == 4) {
      float e0[8], e1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { e0[i] = ylo[i]; e1[i] = yhi[i]; }
      qn[0] = NibbleDot<T>::prep(e0);
      qn[1] = NibbleDot<T>::prep(e1);
    }
  }
  const float rsqrt_d = 1.0f / p.sqrt_d;
  // stash / MSB-logit stores without branches: buffer stores whose descriptor ends at row `hi` (a row past it is dropped by
  // the bounds check) and whose per-lane offset is out of range for every lane but the row's first (c != 0)
  T* stashp = p.scores ? p.scores + b * p.sc_sb + h * p.sc_sh : nullptr;
  const __amdgpu_buffer_rsrc_t rs_st = make_rsrc(stashp, stashp ? (int64_t)hi * (int64_t)sizeof(T) : 0);
  const __amdgpu_buffer_rsrc_t rs_lgw = make_rsrc(lgb, PASS == 1 ? (int64_t)hi * 4 : 0);
  const int vo_st = (c == 0) ? r * (UP * (int)sizeof(T)) : 0x40000000, vo_lw = (c == 0) ? r * (UP * 4) : 0x40000000;

  float m_run = -INFINITY, l_run = 0.f, off_run = 0.f;
  float o16[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o16[i] = 0.f;

  if constexpr (app) {
    if (owns_new && tid < LPR) {
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8], va[8], vb[8];
      V8::unpack(nk_raw[0], xlo);
      V8::unpack(nk_raw[1], xhi);
      V8::unpack(nr_raw[0], cc);
      V8::unpack(nr_raw[1], ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
      const int64_t dst = b * p.kv_sb + hkv * p.kv_sh + (int64_t)(N - 1) * D;
      if (p.kc) { V8::stg(p.kc + dst + 8 * c, nk_raw[0]); V8::stg(p.kc + dst + HALF + 8 * c, nk_raw[1]); }
      V8::stg(p.krc + dst + 8 * c, V8::pack(ylo));
      V8::stg(p.krc + dst + HALF + 8 * c, V8::pack(yhi));
      V8::stg(p.vc + dst + 8 * c, nv_raw[0]);
      V8::stg(p.vc + dst + HALF + 8 * c, nv_raw[1]);
      V8::unpack(nv_raw[0], va);
      V8::unpack(nv_raw[1], vb);
      float kx[16], vx[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) { kx[e] = ylo[e]; kx[8 + e] = yhi[e]; vx[e] = va[e]; vx[8 + e] = vb[e]; }
      PlanePieces<D, KB, VB> pp;
      pack_plane_pieces<D, KB, VB>(kx, vx, pp);
      store_plane_words<D, KB, VB>(p.pl, b, hkv, N - 1, c, pp);
      // its MSB logit, exactly as a tile row's
      float a;
      if (KBITS == 4) a = NibbleDot<T>::dot(qn[0], pp.wm[0], 8.f) + NibbleDot<T>::dot(qn[1], pp.wm[KW - 1], 8.f);
      else a = fmaf(-((float)(1 << (KB - 1)) + ((KBITS == 6 && SPATTEN_PQV_MAGIC6) ? kMagic6Hi : 0.f)), qsum, dot_piece<KBITS, KW>(pp.wm, qv));
      a = group_sum<LPR>(a);
      const float sn = a * (pp.ks * (16.f * rsqrt_d));
      if (c == 0) {
        if (stashp) stashp[N - 1] = DT<T>::from_f32(sn);
        lgb[N - 1] = sn;
      }
      m_run = sn; l_run = 1.f;                           // the first key of this thread-row's softmax
      const float wgt = pp.vs;
      fma_piece<VB, VW>(o16, pp.wv, wgt);
      off_run = wgt;
    }
  }
  // every row-group of a tile is computed unconditionally (rows past the split read as zeros); `valid` gates the results
  auto process_tile = [&](Tile& tl, int t0) {
    float sc[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      float a;
      if (KBITS == 4) {      // fields hold msb + 8 (pass 1) / the LSB nibble (pass 2)
        a = NibbleDot<T>::dot(qn[0], tl.kw[u][0], PASS == 1 ? 8.f : 0.f) + NibbleDot<T>::dot(qn[1], tl.kw[u][KW - 1], PASS == 1 ? 8.f : 0.f);
      } else {               // fields hold msb + 2^(KB-1)
        // (an explicit fma: the appended row's logit is computed by the same expression elsewhere and must round the same way)
        a = fmaf(-((float)(1 << (KB - 1)) + ((KBITS == 6 && SPATTEN_PQV_MAGIC6) ? kMagic6Hi : 0.f)), qsum, dot_piece<KBITS, KW>(tl.kw[u], qv));
      }
      a = group_sum<LPR>(a);
      const float s = a * (__uint_as_float(tl.ks[u]) * (PASS == 1 ? 16.f * rsqrt_d : rsqrt_d));
      sc[u] = (PASS == 2) ? s + __uint_as_float(tl.lg[u]) : s;
    }
    // stash (:116-119) and the MSB logits: a lane's UP rows are consecutive — one store each when the whole tile lies inside the
    // split (wave-uniform test), else row by row against the descriptor's end
    if (t0 + TILE <= hi) {
      uint32_t lw[UP];
#pragma unroll
      for (int u = 0; u < UP; ++u) lw[u] = __float_as_uint(sc[u]);
      if constexpr (sizeof(T) == 4) {
        if (stashp) bstore<UP>(rs_st, vo_st, t0 * 4, lw);
      } else {
        uint32_t sw[UP / 2];
#pragma unroll
        for (int u = 0; u < UP; u += 2) {
          const T a0 = DT<T>::from_f32(sc[u]), a1 = DT<T>::from_f32(sc[u + 1]);
          sw[u / 2] = (uint32_t)*reinterpret_cast<const unsigned short*>(&a0) | ((uint32_t)*reinterpret_cast<const unsigned short*>(&a1) << 16);
        }
        bstore<UP / 2>(rs_st, vo_st, t0 * 2, sw);
      }
      if (PASS == 1) bstore<UP>(rs_lgw, vo_lw, t0 * 4, lw);
    } else {
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const float s = sc[u];
        if constexpr (sizeof(T) == 4) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), rs_st, vo_st, (t0 + u) * 4, 0);
        } else {
          const T sv = DT<T>::from_f32(s);
          __builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<const unsigned short*>(&sv), rs_st, vo_st, (t0 + u) * 2, 0);
        }
        if (PASS == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), rs_lgw, vo_lw, (t0 + u) * 4, 0);
      }
    }
    float m_new = m_run;
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int j = t0 + r * UP + u;
      const bool valid = j < hi;
      const float s = valid ? sc[u] : -INFINITY;
      sc[u] = s;
      m_new = fmaxf(m_new, s);
    }
    if (m_new > m_run) {
      const float alpha = __expf(m_run - m_new);
      l_run *= alpha;
      off_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) o16[i] *= alpha;
      m_run = m_new;
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const float pj = __expf(sc[u] - m_run);      // exp(-inf) = 0; m_run is finite once a valid row has been seen
      const float pz = (sc[u] == -INFINITY) ? 0.f : pj;
      l_run += pz;
      const float wgt = pz * __uint_as_float(tl.vs[u]);
      fma_piece<VB, VW>(o16, tl.vw[u], wgt);
      off_run += wgt;
#ifdef SPATTEN_PQV_PVBAR     // A/B: keep the row-groups' products apart (16 independent accumulators per row-group)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  for (int t0 = lo; t0 < hi; t0 += 2 * TILE) {
    issue(tile_b, t0 + TILE);
    __builtin_amdgcn_sched_barrier(0);
    process_tile(tile_a, t0);
    issue(tile_a, t0 + 2 * TILE);
    __builtin_amdgcn_sched_barrier(0);
    if (t0 + TILE < hi) process_tile(tile_b, t0 + TILE);
  }
  {   // the fields hold qv + 2^(VB-1): take the constant out through the weight sum
    const float k = (float)(1 << (VB - 1)) * off_run;
#pragma unroll
    for (int i = 0; i < 16; ++i) o16[i] -= (VB == 6 && SPATTEN_PQV_MAGIC6) ? fmaf(piece_bias6(i), off_run, k) : k;
  }

  // ---- reconcile the row groups (decode_attn.hip): per-wave max and sums in registers, one LDS hop across the waves ----
  {
    const float mw = wave_max(m_run);
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - mw);
    l_run *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) o16[i] *= alpha;
    m_run = mw;
  }
  if (LPR == 4) {
    l_run += dpp_mov<kDppRor8>(l_run);
    l_run += dpp_mov<kDppRor4>(l_run);
#pragma unroll
    for (int i = 0; i < 16; ++i) { o16[i] += dpp_mov<kDppRor8>(o16[i]); o16[i] += dpp_mov<kDppRor4>(o16[i]); }
  } else {
    l_run += dpp_mov<kDppRor8>(l_run);
#pragma unroll
    for (int i = 0; i < 16; ++i) o16[i] += dpp_mov<kDppRor8>(o16[i]);
  }
  l_run = xor32_sum(xor16_sum(l_run));
#pragma unroll
  for (int i = 0; i < 16; ++i) o16[i] = xor32_sum(xor16_sum(o16[i]));
  if (lane < LPR) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_o[wave][8 * lane + i] = o16[i];
      s_o[wave][HALF + 8 * lane + i] = o16[8 + i];
    }
    if (lane == 0) { s_o[wave][D] = l_run; s_o[wave][D + 1] = m_run; }
  }
  __syncthreads();
  float o_tot = 0.f, l_tot = 0.f;
  {
    if constexpr (kPqvWaves == 4) {
      const float m0 = s_o[0][D + 1], m1 = s_o[1][D + 1], m2 = s_o[2][D + 1], m3 = s_o[3][D + 1];
      const float m_wg = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      const float mu = (m_wg == -INFINITY) ? 0.f : m_wg;
      const float w0 = __expf(m0 - mu), w1 = __expf(m1 - mu), w2 = __expf(m2 - mu), w3 = __expf(m3 - mu);
      if (tid < D) o_tot = (s_o[0][tid] * w0 + s_o[1][tid] * w1) + (s_o[2][tid] * w2 + s_o[3][tid] * w3);
      l_tot = (s_o[0][D] * w0 + s_o[1][D] * w1) + (s_o[2][D] * w2 + s_o[3][D] * w3);
      m_run = m_wg;
    } else {
      float mw[kPqvWaves], m_wg = -INFINITY;
#pragma unroll
      for (int w = 0; w < kPqvWaves; ++w) { mw[w] = s_o[w][D + 1]; m_wg = fmaxf(m_wg, mw[w]); }
      const float mu = (m_wg == -INFINITY) ? 0.f : m_wg;
#pragma unroll
      for (int w = 0; w < kPqvWaves; ++w) {
        const float ww = __expf(mw[w] - mu);
        if (tid < D) o_tot = fmaf(s_o[w][tid], ww, o_tot);
        l_tot = fmaf(s_o[w][D], ww, l_tot);
      }
      m_run = m_wg;
    }
  }
  T* outp = p.out + b * p.out_sb + h * D;
  // head importance: the MSB pass adds only for confident heads, the refetch pass for the heads it recomputes
  auto add_head_abs = [&](float val, bool active, bool commit) {
    float v = wave_sum(active ? fabsf(DT<T>::round(val)) : 0.f);
    __syncthreads();
    if (lane == 0) s_o[0][wave] = v;
    __syncthreads();
    if (tid == 0 && commit) {
      float tot = (s_o[0][0] + s_o[0][1]) + (s_o[0][2] + s_o[0][3]);
#pragma unroll
      for (int w = 4; w < kPqvWaves; ++w) tot += s_o[0][w];
      p.head_abs[unit] += tot;
    }
  };
  if (p.S == 1) {
    if (tid < D) outp[tid] = DT<T>::from_f32(o_tot / l_tot);
    if (p.lse != nullptr && tid == 0) { float* ls = p.lse + (int64_t)unit * 2; ls[0] = m_run; ls[1] = l_tot; }
    const bool need1 = PASS == 1 && (1.0f / l_tot) < p.thr;                              // max prob = exp(0) / sum
    if (PASS == 1 && tid == 0) p.need[unit] = need1 ? 1 : 0;
    if (p.head_abs != nullptr) add_head_abs(o_tot / l_tot, tid < D, !need1);
    return;
  }

  // ---- publish the partial; merge (decode_attn.hip's protocol on the same workspace) -------------------------------------
  unsigned long long* ws = p.ws_part + (int64_t)unit * p.ws_unit;
  unsigned long long* part = ws + (int64_t)split * (D + 2);
  const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;
  if (!p.poll_merge) {
    if (tid == kPqvThreads - 1)
      s_ticket = __hip_atomic_fetch_add(p.ws_cnt + 2 * unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < D && tid < kPqvThreads - kWave) pqv_store_granule(part + tid, o_tot, tag);
  if (tid == D) { pqv_store_granule(part + D, m_run, tag); pqv_store_granule(part + D + 1, l_tot, tag); }
  if (p.poll_merge) {
    if (split != p.S - 1) return;
  } else {
    __syncthreads();
    if (s_ticket != (unsigned)(p.S - 1)) return;
  }
  constexpr int KBm = 8;
  const int Gr = p.S > KBm ? G : 1;
  const int e = tid % D, g = tid / D;
  float mg = -INFINITY, lg = 0.f, og = 0.f;
  bool expired = false;
  if (g < Gr) {
    for (int s0 = g; s0 < p.S; s0 += KBm * Gr) {
      unsigned long long ga[KBm], gm[KBm], gl[KBm];
      int spins = 0;
      bool landed;
      do {
#pragma unroll
        for (int k = 0; k < KBm; ++k) {
          const int sc_ = (s0 + k * Gr) < p.S ? (s0 + k * Gr) : g;
          const unsigned long long* qg = ws + (int64_t)sc_ * (D + 2);
          ga[k] = __hip_atomic_load(qg + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gm[k] = __hip_atomic_load(qg + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gl[k] = __hip_atomic_load(qg + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned diff = 0u;
#pragma unroll
        for (int k = 0; k < KBm; ++k)
          diff |= ((unsigned)(ga[k] >> 32) ^ tag) | ((unsigned)(gm[k] >> 32) ^ tag) | ((unsigned)(gl[k] >> 32) ^ tag);
        landed = diff == 0u;
      } while (!landed && ++spins < (1 << 16));
      expired |= !landed;
      float a[KBm], ms[KBm], ls[KBm];
#pragma unroll
      for (int k = 0; k < KBm; ++k) {
        const bool live = (s0 + k * Gr) < p.S;
        a[k] = live ? __uint_as_float((unsigned)ga[k]) : 0.f;
        ms[k] = live ? __uint_as_float((unsigned)gm[k]) : -INFINITY;
        ls[k] = live ? __uint_as_float((unsigned)gl[k]) : 0.f;
      }
      float mn = mg;
#pragma unroll
      for (int k = 0; k < KBm; ++k) mn = fmaxf(mn, ms[k]);
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu);
      og *= w0; lg *= w0;
#pragma unroll
      for (int k = 0; k < KBm; ++k) {
        const float w = __expf(ms[k] - mu);
        og = fmaf(a[k], w, og);
        lg = fmaf(ls[k], w, lg);
      }
      mg = mn;
    }
  }
  if (expired) {
    atomicOr(p.ws_err, 1u);
    og = __builtin_nanf("");
  }
  if (Gr > 1) {
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
  if (g == 0) outp[e] = DT<T>::from_f32(og / lg);
  if (p.head_abs != nullptr) {
    if (PASS == 1 && tid == 0) s_ticket = (1.0f / lg) < p.thr ? 1u : 0u;
    if (PASS == 1) __syncthreads();
    add_head_abs(og / lg, g == 0, !(PASS == 1 && s_ticket != 0u));
  }
  if (tid == 0) {
    if (p.lse != nullptr) { float* ls = p.lse + (int64_t)unit * 2; ls[0] = mg; ls[1] = lg; }
    if (PASS == 1) p.need[unit] = (1.0f / lg) < p.thr ? 1 : 0;
    p.ws_cnt[2 * unit + 1] = gen + 1u;
    if (!p.poll_merge) __hip_atomic_store(p.ws_cnt + 2 * unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Workgroups of one instantiation the chip holds at once (for the polling merge: the grid must be co-resident by
// construction).  The runtime's occupancy answer, capped at 2 per CU (every instantiation's LDS / SGPR budget admits two;
// the API can overstate beyond that — MI355X_MICROARCH.md "Residency"), asked once per instantiation.
template <typename K> static int resident_capacity(K kernel) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kPqvThreads, 0) != hipSuccess) per_cu = 1;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  return std::max(1, std::min(per_cu, 2)) * cus;
}

template <typename T, int D, int KB, int VB>
static int launch_pqv(PqvParams<T>& p, int n_active, bool dyn, bool msb_only, int env_poll, hipStream_t st) {
  const bool app = p.k_new != nullptr;
  constexpr int UP = SPATTEN_PQV_UP;
  const dim3 grid((unsigned)p.S, (unsigned)n_active, (unsigned)p.B), blk(kPqvThreads);
  // (the pass-1 and pass-2 kernels of a profile have different register footprints: the smaller capacity decides)
  static const int cap_static = std::min(resident_capacity(pqv_decode_kernel<T, D, KB, VB, 1, false, UP>),
                                         resident_capacity(pqv_decode_kernel<T, D, KB, VB, 2, false, UP>));
  static const int cap_dyn = std::min(resident_capacity(pqv_decode_kernel<T, D, KB, VB, 1, true, UP>),
                                      resident_capacity(pqv_decode_kernel<T, D, KB, VB, 2, true, UP>));
  p.poll_merge = (env_poll != 0 && p.S > 1 && (long long)p.S * n_active * p.B <= (dyn ? cap_dyn : cap_static)) ? 1 : 0;
  // ---- the refetch pass (round 6): at the reference traces' rate 2-3 of 30 heads are flagged per layer-step; laid out like pass 1
  // their rows sit on S CUs each and the pass takes as long as if every head were flagged (a CU streams at ~21 B/ns).  So pass 2
  // gets its own, finer split-N (~256-row chunks, <= 64 splits) over a grid whose rows are the RANKS of the flagged heads: a few
  // flagged heads spread over the whole chip, unflagged rows of the grid exit at once.  The merge polls: workgroups are dispatched
  // in order, a head's merging (last) split after its siblings, and no other workgroup ever waits — safe beyond residency.
  PqvParams<T> p2 = p;
  static int env_c = -1;
  if (env_c < 0) { const char* e = getenv("SPATTEN_PQV_COMPACT"); env_c = e ? atoi(e) : 1; }
  if (env_c && !msb_only && n_active <= 64) {
    const int lay = p.S * p.chunk;                       // the layout length pass 1 was laid out for
    const int ws_splits = (int)(p.ws_unit / (D + 2));
    const int rows2 = env_c >= 64 ? env_c : 256;          // (SPATTEN_PQV_COMPACT=rows: A/B of the chunk length)
    int S2 = std::min({kDecodeMaxSplits, ws_splits, std::max(p.S, ceil_div(lay, rows2))});
    const int chunk2 = ceil_div(ceil_div(lay, S2), 8) * 8;
    S2 = ceil_div(lay, chunk2);
    p2.S = S2; p2.chunk = chunk2; p2.compact = 1; p2.n_act = n_active; p2.S1 = p.S; p2.chunk1 = p.chunk;
    p2.poll_merge = (env_poll != 0 && S2 > 1) ? 1 : 0;
  }
  const dim3 grid2((unsigned)p2.S, (unsigned)n_active, (unsigned)p.B);
  if (dyn) {
    if (app) hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 1, true, UP, true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 1, true, UP>), grid, blk, 0, st, p);
    if (!msb_only) hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 2, true, UP>), grid2, blk, 0, st, p2);
  } else {
    if (app) hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 1, false, UP, true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 1, false, UP>), grid, blk, 0, st, p);
    if (!msb_only) hipLaunchKernelGGL((pqv_decode_kernel<T, D, KB, VB, 2, false, UP>), grid2, blk, 0, st, p2);
  }
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

template <typename T>
static int run_pqv(const spatten_pq_decode_args_t* a, const PlanesDev& pd, hipStream_t st) {
  const int kb = a->planes->key_msb_bits;      // (the value width follows: supported profiles are (4, 8), (8, 8), (6, 6))
  const int d = a->head_dim;
  const int n_active = a->head_ids ? a->n_active_heads : a->heads;
  const int units = a->batch * a->heads;
  const int ws_splits = a->workspace_splits > 0 ? a->workspace_splits : kDecodeMaxSplits;
  const int lay = (!a->step_state && a->kv_len_layout > a->kv_len) ? a->kv_len_layout : a->kv_len;
  int S = a->n_splits > 0 ? a->n_splits : decode_auto_splits(a->batch * n_active, d, lay, 2, false);
  if (S > lay) S = lay;
  if (S > kDecodeMaxSplits) S = kDecodeMaxSplits;
  const int chunk = ceil_div(ceil_div(lay, S), 8) * 8;
  S = ceil_div(lay, chunk);
  if (S > 1 && (!a->workspace || S > ws_splits)) return SPATTEN_ERR_INVALID;
  static int env_poll = -1;
  if (env_poll < 0) { const char* e = getenv("SPATTEN_DECODE_POLL"); env_poll = e ? atoi(e) : 1; }
  PqvParams<T> p;
  p.q = (const T*)a->q; p.q_sb = a->q_sb; p.q_sh = a->q_sh;
  p.pl = pd;
  p.cos = (const T*)a->cos; p.sin = (const T*)a->sin; p.table_rows = a->table_rows;
  p.step = (const int32_t*)a->step_state;
  p.pos_q = a->pos_q;
  if (a->step_state) {   // the state's staged rotary rows stand in for the table: row 0 = the query's position
    p.cos = (const T*)((const char*)a->step_state + kStepHeader);
    p.sin = p.cos + 2 * (d / 2); p.table_rows = 2; p.pos_q = 0;
  }
  p.out = (T*)a->out; p.out_sb = a->out_sb;
  p.scores = (T*)a->scores; p.sc_sb = a->sc_sb; p.sc_sh = a->sc_sh;
  p.lse = a->lse; p.head_ids = a->head_ids;
  p.thr = a->threshold; p.need = a->need_lsb; p.head_abs = a->head_abs_acc;
  const size_t cnt_bytes = decode_cnt_bytes((size_t)units);
  p.ws_err = (unsigned*)a->workspace;
  p.ws_cnt = a->workspace ? (unsigned*)((char*)a->workspace + kDecodeWsHeader) : (unsigned*)a->cos;
  p.ws_part = a->workspace ? (unsigned long long*)((char*)a->workspace + kDecodeWsHeader + cnt_bytes) : nullptr;
  p.ws_unit = (int64_t)ws_splits * (d + 2);
  p.B = a->batch; p.H = a->heads; p.Hkv = a->kv_heads; p.N = a->kv_len; p.S = S; p.chunk = chunk;
  p.poll_merge = 0;      // decided per instantiation in launch_pqv (co-residency of the whole grid)
  p.compact = 0; p.n_act = n_active; p.S1 = S; p.chunk1 = chunk;
  p.sqrt_d = sqrtf((float)d);
  p.k_new = (const T*)a->k_new; p.v_new = (const T*)a->v_new; p.new_sb = a->new_sb; p.new_sh = a->new_sh;
  p.kc = (T*)a->k_cache; p.krc = (T*)a->kr_cache; p.vc = (T*)a->v_cache; p.kv_sb = a->kv_sb; p.kv_sh = a->kv_sh;
  p.nr_row = a->step_state ? 1 : std::min(a->kv_len - 1, a->table_rows - 1);
  const bool dyn = a->step_state != nullptr, msb_only = (a->flags & SPATTEN_PQ_MSB_PASS_ONLY) != 0;
#define SPATTEN_PQV(DD, KB, VB) return launch_pqv<T, DD, KB, VB>(p, n_active, dyn, msb_only, env_poll, st)
  if (d == 128) {
    if (kb == 4) SPATTEN_PQV(128, 4, 8);
    if (kb == 8) SPATTEN_PQV(128, 8, 8);
    SPATTEN_PQV(128, 6, 6);
  }
  if (kb == 4) SPATTEN_PQV(64, 4, 8);
  if (kb == 8) SPATTEN_PQV(64, 8, 8);
  SPATTEN_PQV(64, 6, 6);
#undef SPATTEN_PQV
}

}  // namespace spatten

using namespace spatten;

extern "C" int spatten_attn_decode_pq(const spatten_pq_decode_args_t* a, void* stream) {
  if (!a || a->struct_size != sizeof(spatten_pq_decode_args_t)) return SPATTEN_ERR_INVALID;
  PlanesDev pd;
  if (!a->q || !a->out || !a->cos || !a->sin || !a->need_lsb || !planes_to_dev(a->planes, pd)) return SPATTEN_ERR_INVALID;
  if (a->batch <= 0 || a->heads <= 0 || a->kv_heads <= 0 || a->heads % a->kv_heads != 0 || a->kv_len <= 0) return SPATTEN_ERR_INVALID;
  if (!a->step_state && (a->pos_q < 0 || a->pos_q >= a->table_rows)) return SPATTEN_ERR_INVALID;
  if (a->kv_len_layout < 0) return SPATTEN_ERR_INVALID;
  if ((a->k_new == nullptr) != (a->v_new == nullptr)) return SPATTEN_ERR_INVALID;
  if (a->k_new && (!a->kr_cache || !a->v_cache)) return SPATTEN_ERR_INVALID;
  const int n_active = a->head_ids ? a->n_active_heads : a->heads;
  if (n_active <= 0 || n_active > a->heads) return SPATTEN_ERR_INVALID;
  if (!ok_dtype(a->dtype)) return SPATTEN_ERR_INVALID;
  if (a->head_dim != 64 && a->head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  if (!pq_profile_supported(a->planes->key_msb_bits, a->planes->value_bits)) return SPATTEN_ERR_UNSUPPORTED;
  if (a->dtype == SPATTEN_BF16) return run_pqv<bf16_t>(a, pd, (hipStream_t)stream);
  if (a->dtype == SPATTEN_F16) return run_pqv<f16_t>(a, pd, (hipStream_t)stream);
  return run_pqv<float>(a, pd, (hipStream_t)stream);
}
