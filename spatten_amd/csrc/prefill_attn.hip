// prefill_attn.hip — attention for a block of queries (modify_llama.py:86-147 at q_len > 1).
//
// Three legs behind one entry point (spatten_attn_prefill):
//   * rows leg  (fp32, very short q, head_dim 256): the decode kernel, one softmax row per workgroup column
//     (decode_rows in decode_attn.hip) — exact fp32, no matrix cores needed for a handful of rows.
//   * flash leg (bf16/f16, head_dim 64/128): compute-bound -> MFMA.  One preparation kernel and the flash kernel on
//     the same stream:
//       (1) the queries are rotated in the flash kernel's prologue (reference rounding, modify_llama.py:92);  K is NOT
//           rotated here: the cache already carries the rotated shadow Kr (see decode_attn.hip),
//       (2) vt kernel: V [keys][d] -> Vt [d][keys] scratch so that the P·V matrix product finds its
//           contraction index (keys) contiguous per lane; inside every 32-key block the keys are permuted
//           into the order in which a lane holds them after the Q·K^T product (no shuffles between the two
//           products),
//       (3) flash kernel: workgroup = 256 queries (8 waves x 32, one workgroup per CU), key tiles staged in LDS
//           (XOR-swizzled 16-byte slots, conflict-free ds_read_b128), S^T = Kr·Qrot^T and O^T = Vt·P^T on
//           v_mfma_f32_32x32x16_{bf16,f16}: a lane owns ONE query column, so the online softmax needs a
//           single cross-lane exchange (lane <-> lane+32) per tile.  The two halves of the workgroup run one phase
//           apart (matrix phase beside vector phase on every SIMD): (3b) 64-key tiles staged through registers,
//           (3c) 128-key tiles brought in by LDS-DMA.
//     The stash (pre-mask logits, modify_llama.py:116-119) and the column importance (kv_cache_token_pruning.py:51
//     includes the acausal logits) are optional by-products; when requested, key tiles above the causal
//     diagonal are still scored (but skip softmax / P·V).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mfma_tiles.h"

namespace spatten {

#ifdef SPATTEN_PF_TRACE   // developer instrumentation (tools/probe_pf_trace.py): phase timestamps of workgroup 0
__device__ unsigned long long* g_pf_trace = nullptr;
#define PF_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (g_pf_trace && blockIdx.x == 0 && lane == 0 && t >= 8 && t < 24)                                  \
      g_pf_trace[((wave * 16 + (t - 8)) * 8) + (slot)] = __builtin_readcyclecounter();                   \
  } while (0)
#else
#define PF_STAMP(slot)
#endif



// position of key (0..31, within its 32-key block) in a Vt row: the order in which the Q·K^T accumulator
// registers of a lane enumerate keys:  key = (r&3) + 8*(r>>2) + 4*hi  for register r (0..15), half hi.
__host__ __device__ inline int vt_key_of_pos(int pos) {
  const int t = pos >> 4, hi = (pos >> 3) & 1, e = pos & 7, r = t * 8 + e;
  return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

// ------------------------------------------------------------------------------------------------
// (2) V [B,Hkv,N,d] -> Vt [B,Hkv,d,Npad]  (16-bit dtypes; zero beyond N)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void vt_kernel(const uint16_t* __restrict__ v, int64_t kv_sb, int64_t kv_sh,
                                                 uint16_t* __restrict__ vt, int N, int Npad, int Hkv) {
  constexpr int PITCH = D + 8;
  __shared__ __attribute__((aligned(16))) uint16_t tile[64 * PITCH];
  const int tid = threadIdx.x, tile_i = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const uint16_t* src = v + b * kv_sb + hkv * kv_sh;
  for (int id = tid; id < 64 * (D / 8); id += 256) {
    const int key = id / (D / 8), c8 = id % (D / 8);
    const int j = tile_i * 64 + key;
    u32x4 x = {0u, 0u, 0u, 0u};
    if (j < N) x = *reinterpret_cast<const u32x4*>(src + (int64_t)j * D + c8 * 8);
    *reinterpret_cast<u32x4*>(&tile[key * PITCH + c8 * 8]) = x;
  }
  __syncthreads();
  uint16_t* dst = vt + ((int64_t)(b * Hkv + hkv) * D) * Npad + (int64_t)tile_i * 64;
  for (int id = tid; id < D * 8; id += 256) {
    const int dv = id >> 3, pc = id & 7;
    uint32_t w[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int p0 = pc * 8 + 2 * e2, p1 = p0 + 1;
      const int k0 = (p0 & 32) + vt_key_of_pos(p0 & 31), k1 = (p1 & 32) + vt_key_of_pos(p1 & 31);
      w[e2] = (uint32_t)tile[k0 * PITCH + dv] | ((uint32_t)tile[k1 * PITCH + dv] << 16);
    }
    u32x4 o = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4*>(dst + (int64_t)dv * Npad + pc * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// (3) flash kernel
// ------------------------------------------------------------------------------------------------
template <typename T>
struct FlashParams {
  const T* q; int64_t q_sb, q_sh, q_sq;   // un-rotated queries [B,H,q_len,D] (any strides, d contiguous)
  const T* cos; const T* sin; int table_rows;             // rotary half tables [rows, D/2]
  const int64_t* pos_ids; int64_t pos_sb; int pos_q0;     // query positions: pos_ids[b][i] or pos_q0 + i
  const T* kr;        // rotated shadow [B,Hkv,cap,D]
  int64_t kv_sb, kv_sh;
  const T* vt;        // [B,Hkv,D,Npad]
  const T* v; int64_t v_sb, v_sh;   // the value cache itself [B,Hkv,>=N,D] (rows contiguous): read by the transposing
                                    // LDS reads of the 128-key kernel at D = 128 (SPATTEN_PF_VTR), no Vt pre-pass
  const T* mask; int64_t mask_sb, mask_sq;
  T* out; int64_t out_sb, out_sq;
  T* scores; int64_t sc_sb, sc_sh, sc_sq;
  float* col_imp;     // [B,H,N]
  float* lse;         // optional [B,H,q_len,2]: (reference max m, sum exp(s - m)) of every query row — what
                      // spatten_importance_accumulate_prefill turns into softmax probabilities without a stash
  // progressive-quant keys (prefill_pp128_kernel<.., PQK>): kr holds INTEGER-valued keys (msb*16 or the full q8, exact in
  // the 16-bit dtype), kscale the per-key factor scale / sqrt(d) applied to the fp32 score
  const float* kscale; int64_t ks_sb, ks_sh;   // [B,Hkv,N] fp32
  int32_t* need;      // [B,H,q_len]: pass 1 writes max prob < thr, pass 2 recomputes the flagged rows
  float pq_thr;
  // pass 2 (round 5): the flagged rows of a (b, h), compacted — ascending query indices [B,H,q_len] and their count [B,H]
  // (pq_rows_compact_kernel).  A workgroup of pass 2 serves 256 LIST entries, not 256 consecutive rows: at a realistic refetch
  // rate (5 % of the rows: nearly every 256-row block and 4 of 5 waves hold a flagged row) the pass costs what it recomputes
  const int32_t* rows; const int32_t* row_cnt;
  int B, H, Hkv, q_len, N, Npad, causal, nqb;
  int pair;           // prefill_pp128_kernel<..., PAIR>: a workgroup's halves take the 128-row blocks i and n - 1 - i (see the kernel)
  int vtr;     // this launch reads V through the transposing LDS reads (no Vt copy was made)
  int fast;    // SPATTEN_PREFILL_FAST_NUMERICS: fp32 logits, no reference roundings (plain causal / unmasked flash leg only)
  float sqrt_d;
  // key split (prefill_pp128_kernel, plain keys): a (b, h, query block) is served by ksplit workgroups, each over a
  // contiguous range of key tiles; they leave un-normalised partial outputs + (m, l) here and prefill_merge_kernel
  // folds them.  Short query blocks on a long cache (the turn prefill of the multi-turn protocol: 64-500 new tokens on
  // 2048 cached rows) otherwise run on H workgroups of a 256-CU chip.
  int ksplit;
  float* part_o;      // [B*H*nqb][ksplit][256][D] fp32
  float* part_ml;     // [B*H*nqb][ksplit][256][2]
};

// A/B switches (tools/mb/pf_exp.sh rebuilds with -D<macro>=<v>; findings in DESIGN §3.4 and HISTORY.md).
#ifndef SPATTEN_PF_PRIO         // static s_setprio for one half: MEASURED SLOWER (775-789 vs 812 TFLOP/s).  Off.
#define SPATTEN_PF_PRIO 0
#endif
#ifndef SPATTEN_PF_EXPMODE      // harness bits: 1 = FASTNUM; 2 = no DMA in the tile loop and 8 = no softmax (both give WRONG
#define SPATTEN_PF_EXPMODE 0    // results: anatomy only); 4 = DMA_MODE 2; 16 = no exp2 in the softmax (wrong results)
#endif
#ifndef SPATTEN_PF_DMA_MODE     // who issues a stage's LDS-DMA: 0 half 0 in its matrix / half 1 in its vector phase (r01, 762);
#define SPATTEN_PF_DMA_MODE 2   // 1 both in their matrix phase (r02: 810 against 743 for 2); 2 both in their vector phase — r06, after the r05 softmax
                                // diet: 2 is +1.5-2.4 % in both numerics (four alternations on two boxes: 882 / 991-1,022 against 870 / 973-998 TFLOP/s);
                                // 3 = K pieces from the matrix, V pieces from the vector phase: as 1
#endif
// FAST (template flag of prefill_pp128_kernel; run-time opt-in SPATTEN_PREFILL_FAST_NUMERICS of spatten_attn_prefill):
// logits kept in fp32 — no reference roundings (matmul -> dtype, / sqrt(d) -> dtype) — with the scale folded into the
// exponent: -384 of ~609 VALU per wave-tile.  NOT the default: the flash kernel's logits stay the reference's (DESIGN
// §3.4).  The harness bit SPATTEN_PF_EXPMODE & 1 forces it for every launch (anatomy builds).
// VTRP (template flag of prefill_pp128_kernel, d = 128): the V operand of P.V straight from the value rows — the tile is
// DMA-ed row-major ([128 keys][D]) and read with ds_read_b64_tr_b16 (gfx950's transposing LDS read), so the launch needs no
// key-contiguous copy of V (the vt_kernel pre-pass: 9 of the 36 us of the 64-token turn prefill).  Two 8-byte transposing
// reads per fragment instead of one 16-byte read make the flash kernel itself slower, so the form is chosen per shape
// (use_vtr below); SPATTEN_PREFILL_VTR=0|1 forces it (A/B).
#ifndef SPATTEN_PF_VTR_MAXQ
#define SPATTEN_PF_VTR_MAXQ 512  // transposing reads for query blocks up to this length
#endif
#ifndef SPATTEN_PF_LOCKSTEP      // A/B: both halves in the same phase (no ping-pong offset) — with DMA_MODE 1 both issue their pieces of
#define SPATTEN_PF_LOCKSTEP 0    // stage t+1 at the top of the matrix phase and wait for them at the end of the vector phase
#endif
#ifndef SPATTEN_PF_P1            // the reference roundings of S(t+1) behind the wave's own P.V MFMAs (prefill_pp128_kernel)
#define SPATTEN_PF_P1 0
#endif
#ifndef SPATTEN_PF_ROWSUM_MFMA  // row sums of P on the matrix pipe instead of 64 VALU adds per lane and tile: MEASURED SLOWER
#define SPATTEN_PF_ROWSUM_MFMA 0   // (713 vs 768): the 8 extra MFMAs per tile cost more than the adds they replace.  Off.
#endif
#ifndef SPATTEN_PF_DIET         // r05 A/B switches of prefill_pp128_kernel's softmax (bits; prebuilt variants: tools/mb/build_variant.sh):
#define SPATTEN_PF_DIET 0       //   2 = no per-tile maximum on fully visible tiles (exponentials against the running maximum, the lane's
#endif                          //       sum bounds every one of them; the rare tile that outgrows it is redone): parity-green, MEASURED
                                //       NEUTRAL-TO-SLOWER (824-831 vs 841-850 TFLOP/s at q = N = 8192, alternating in one call) — off;
                                //   4 = row sums of the ROUNDED P on the packed-dot unit: slower (785; v_dot2 costs ~2.4 plain issues)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kDeferMax = 8.0f;   // natural-log units of the scaled logits
constexpr float kSumBound = 2980.0f;   // < e^kDeferMax: a lane whose 64 exponentials sum to no more holds none above e^kDeferMax

template <int ROWB> __device__ inline int lds_off(int row, int slot) {
  // 16-byte slots, XOR-swizzled so the 16 lanes of a ds_read_b128 group land on distinct bank quads
  if (ROWB == 256) return row * 256 + ((slot ^ (row & 15)) << 4);
  return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
}

// ------------------------------------------------------------------------------------------------
// (3b) ping-pong flash kernel: 8 waves = 256 queries per workgroup, one workgroup per CU, two waves per SIMD.
// PMC of this round's first kernel (4 waves x 32 queries, two workgroups per CU; tools/pmc_prefill.sh): the two waves
// that share a SIMD run the same code in phase — both in their MFMA stretch (the matrix pipe alternates between them),
// then both in their softmax stretch (the matrix pipe idles) — so the SIMD's time is the SUM of the two streams
// (MFMA pipe 33 % busy, issue port 73 %).
// Here the two halves of the workgroup (waves 0-3 / 4-7, one of each on every SIMD) are held exactly one phase
// apart by the workgroup barrier: while one half runs its matrix phase  { O += Vt(t)·P(t) ; S = K(t+1)·Q }
// the other runs its vector phase  { publish staged K/V pieces to LDS, issue the next global loads, softmax(S) -> P },
// then they swap.  Matrix beside vector on every SIMD, by construction.
//
// LDS ring: stage j = { K(j+1), Vt(j) } lives in slot j & 1 and is read in the matrix phase of iteration j — by
// half 0 in global phase 2j, by half 1 in 2j+1.  Stage j+1 is written during phases 2j (half 1's pieces) and 2j+1
// (half 0's pieces): its slot last held stage j-1, read for the last time in phase 2j-1.
// ------------------------------------------------------------------------------------------------
template <typename T, int D, bool STASH, bool COLIMP, bool MASK>
__global__ __launch_bounds__(512, 1) void prefill_pp_kernel(const FlashParams<T> p) {
  constexpr int KK = D / 16, DB = D / 32, KROWB = D * 2;
  constexpr int KBYTES = 64 * KROWB, VBYTES = D * 128, BUF = KBYTES + VBYTES;
  constexpr int KPC = 64 * (D / 8) / 512, VPC = D * 8 / 512;   // 16-byte pieces per thread per tile (K, Vt)
  using frag = typename Mfma<T>::frag;
  constexpr int SPITCH = 80;
  constexpr int SBYTES = STASH ? 8 * 32 * SPITCH : 0;
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF + SBYTES];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, qi = lane & 31, hi = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);          // which half of the workgroup (wave-uniform)
  const int nqb = p.nqb;                                             // 256-query blocks per head
  int h, qblk, b;
  // XCD-aware work order.  Workgroup i runs on XCD i % 8 (observed dispatch; used for speed only) and every query block
  // of a head re-reads that head's K / Vt tiles, so the blocks of one head are kept on ONE XCD at a time: its private
  // 4 MiB L2 then holds exactly the 2 x 2 MiB a head needs at N = 8192 instead of thrashing over 8 heads.  Within a
  // head: longest (latest) query blocks first.
  {
    const int i = blockIdx.x;
    const int per_b = p.H * nqb;
    b = i / per_b;
    const int j = i - b * per_b;
    if ((p.H & 7) == 0) {
      const int xx = j & 7, ss = j >> 3;
      h = xx + 8 * (ss / nqb);
      qblk = nqb - 1 - (ss % nqb);
    } else {
      h = j / nqb;
      qblk = nqb - 1 - (j % nqb);
    }
  }
  const int hkv = p.Hkv == p.H ? h : h / (p.H / p.Hkv);
  const int q0 = qblk * 256 + wave * 32;
  const int myq = q0 + qi;
  const bool qvalid = myq < p.q_len;
  const int P = p.N - p.q_len;
  const float rsqrt_d = 1.0f / p.sqrt_d;

  // Q fragments, rotated here (modify_llama.py:92, the reference's three rounded ops): fragment kk holds elements
  // [16kk + 8hi, +8) of the row, so kk and kk + KK/2 are exactly the (x[i], x[i + d/2]) pairs RoPE combines
  frag qf[KK];
  {
    const int qq = min(myq, p.q_len - 1);
    const T* qrow = p.q + b * p.q_sb + h * p.q_sh + (int64_t)qq * p.q_sq;
    int ps = p.pos_ids ? (int)p.pos_ids[b * p.pos_sb + qq] : p.pos_q0 + qq;
    ps = min(max(ps, 0), p.table_rows - 1);
    const T* cr = p.cos + (int64_t)ps * (D / 2);
    const T* sr = p.sin + (int64_t)ps * (D / 2);
#pragma unroll
    for (int kk = 0; kk < KK / 2; ++kk) {
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + 16 * kk + 8 * hi), xlo);
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + D / 2 + 16 * kk + 8 * hi), xhi);
      Vec8<T>::unpack(Vec8<T>::ldg(cr + 16 * kk + 8 * hi), cc);
      Vec8<T>::unpack(Vec8<T>::ldg(sr + 16 * kk + 8 * hi), ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[kk][e] = DT<T>::from_f32(ylo[e]); qf[kk + KK / 2][e] = DT<T>::from_f32(yhi[e]); }
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int wg_q_end = min(p.q_len, qblk * 256 + 256);
  const int att_keys = p.causal ? min(p.N, P + wg_q_end) : p.N;
  const int n_att_tiles = (att_keys + 63) / 64;                          // workgroup-wide
  const int n_tiles = (STASH || COLIMP) ? (p.N + 63) / 64 : n_att_tiles;
  const int my_vis = p.causal ? min(p.N, P + myq + 1) : p.N;
  const int wave_full_keys = p.causal ? min(p.N, P + q0 + 1) : p.N;
  // tiles in which this WAVE has a visible key (its last query sees keys [0, P + q0 + 32)); later tiles only matter
  // to it for the stash / column sums
  const int wave_att_tiles = p.causal ? min(n_att_tiles, (max(min(p.N, P + min(q0 + 32, p.q_len)), 0) + 63) / 64) : n_att_tiles;
  const int wave_tiles = (STASH || COLIMP) ? n_tiles : wave_att_tiles;

  const T* krb = p.kr + b * p.kv_sb + hkv * p.kv_sh;
  const T* vtb = p.vt + ((int64_t)(b * p.Hkv + hkv) * D) * p.Npad;
  const T* maskrow = MASK ? p.mask + b * p.mask_sb + (int64_t)min(myq, p.q_len - 1) * p.mask_sq : nullptr;
  T* stashrow = STASH ? p.scores + b * p.sc_sb + h * p.sc_sh + (int64_t)min(myq, p.q_len - 1) * p.sc_sq : nullptr;
  float* colrow = COLIMP ? p.col_imp + (int64_t)(b * p.H + h) * p.N : nullptr;
  const bool stash_vec = STASH && ((p.sc_sq | p.sc_sh | p.sc_sb) % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.scores) & 15) == 0);
  // COLIMP: B operand of the transposing product — perm[t][k][n] = 1 where column n is the key that the A operand's
  // k-th element of chunk t holds (a lane's registers enumerate keys as (e&3) + 8(e>>2) + 16t + 4hi)
  frag perm[2];
  if (COLIMP) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) perm[t][e] = DT<T>::from_f32(qi == (e & 3) + 8 * (e >> 2) + 16 * t + 4 * hi ? 1.f : 0.f);
  }

  auto k_area = [&](int stage) -> char* { return lds + (stage & 1) * BUF; };            // holds K(stage + 1)
  auto v_area = [&](int stage) -> char* { return lds + (stage & 1) * BUF + KBYTES; };   // holds Vt(stage)
  u32x4 kreg[KPC], vreg[VPC];
  auto load_k = [&](int tile) {
#pragma unroll
    for (int i = 0; i < KPC; ++i) {
      const int id = tid + 512 * i, row = id / (D / 8), slot = id % (D / 8);
      const int j = min(tile * 64 + row, p.N - 1);
      kreg[i] = *reinterpret_cast<const u32x4*>(krb + (int64_t)j * D + slot * 8);
    }
  };
  auto load_v = [&](int tile) {
#pragma unroll
    for (int i = 0; i < VPC; ++i) {
      const int id = tid + 512 * i, dv = id >> 3, slot = id & 7;
      vreg[i] = *reinterpret_cast<const u32x4*>(vtb + (int64_t)dv * p.Npad + tile * 64 + slot * 8);
    }
  };
  auto write_k = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < KPC; ++i) {
      const int id = tid + 512 * i, row = id / (D / 8), slot = id % (D / 8);
      *reinterpret_cast<u32x4*>(buf + lds_off<KROWB>(row, slot)) = kreg[i];
    }
  };
  auto write_v = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < VPC; ++i) {
      const int id = tid + 512 * i, dv = id >> 3, slot = id & 7;
      *reinterpret_cast<u32x4*>(buf + lds_off<128>(dv, slot)) = vreg[i];
    }
  };
  // stage j = { K(j+1), Vt(j) }: which of the two exist
  auto stage_has_k = [&](int j) { return j + 1 < n_tiles; };
  auto stage_has_v = [&](int j) { return j < n_att_tiles; };
  auto load_stage = [&](int j) {
    if (stage_has_k(j)) load_k(j + 1);
    if (stage_has_v(j)) load_v(j);
  };
  auto write_stage = [&](int j) {
    if (stage_has_k(j)) write_k(k_area(j));
    if (stage_has_v(j)) write_v(v_area(j));
  };

  f32x16 s[2];
  frag pf[2][2];
  // S^T = K · Q^T for one 64-key tile: A fragments through a 4-deep register ring (LDS latency ~ 3 MFMAs)
  constexpr int RING = 8;      // A fragments in flight (LDS latency under load ~ several MFMA durations)
  auto qk = [&](const char* kbuf) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
    frag a[RING];
    auto kfrag = [&](int i) { return *reinterpret_cast<const frag*>(kbuf + lds_off<KROWB>((i & 1) * 32 + qi, 2 * (i >> 1) + hi)); };
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = kfrag(i);
#pragma unroll
    for (int i = 0; i < 2 * KK; ++i) {
      s[i & 1] = Mfma<T>::mma(a[i % RING], qf[i >> 1], s[i & 1]);
      if (i + RING < 2 * KK) a[i % RING] = kfrag(i + RING);
    }
    // keep the ring as written: RING reads up front, then one read behind every MFMA (the scheduler otherwise sinks
    // each read to just before its use and the phase becomes LDS-latency-bound)
    __builtin_amdgcn_sched_group_barrier(0x100, RING, 0);
#pragma unroll
    for (int i = 0; i < 2 * KK - RING; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, RING, 0);
  };
  // O^T += Vt · P^T
  auto pv = [&](const char* vbuf) {
    frag a[RING];
    auto vfrag = [&](int i) {   // step i: (kb, t) = i / DB, db = i % DB
      const int db = i % DB, kt = i / DB, kb = kt >> 1, t = kt & 1;
      return *reinterpret_cast<const frag*>(vbuf + lds_off<128>(db * 32 + qi, kb * 4 + t * 2 + hi));
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = vfrag(i);
#pragma unroll
    for (int i = 0; i < 4 * DB; ++i) {
      const int db = i % DB, kt = i / DB;
      o[db] = Mfma<T>::mma(a[i % RING], pf[kt >> 1][kt & 1], o[db]);
      if (i + RING < 4 * DB) a[i % RING] = vfrag(i + RING);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, RING, 0);
#pragma unroll
    for (int i = 0; i < 4 * DB - RING; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, RING, 0);
  };

  // softmax of the scores in s for `tile` -> pf (and the optional by-products); updates m_run / l_run / o
  auto softmax_tile = [&](int tile) {
    const bool attend = tile < wave_att_tiles;
    const bool edge = tile * 64 + 64 > wave_full_keys;
    // (1) both reference roundings of every logit (matmul -> dtype, "/ sqrt(d)" -> dtype, modify_llama.py:111-113),
    //     two scores at a time; afterwards s holds exact model-dtype values
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 x = round2<T>(f32x2{s[kb][r], s[kb][r + 1]});
        const f32x2 v = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
        s[kb][r] = v[0];
        s[kb][r + 1] = v[1];
      }
    // (2) by-products, from the pre-mask logits
    if (STASH) {                                                                             // :116-119
      if (stash_vec) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          // registers 4g..4g+3 hold 4 consecutive keys -> one 8-byte LDS store per group; then the wave's 32 queries x
          // 32 keys leave as 64-byte row segments, 16 B per lane
          char* sw = lds + 2 * BUF + wave * (32 * SPITCH);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            T q4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q4[e] = DT<T>::from_f32(s[kb][4 * g + e]);
            *reinterpret_cast<u32x2*>(sw + qi * SPITCH + 8 * hi + 16 * g) = *reinterpret_cast<u32x2*>(q4);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int id = lane + 64 * i, row = id >> 2, c4 = id & 3;
            const u32x4 piece = *reinterpret_cast<const u32x4*>(sw + row * SPITCH + c4 * 16);
            const int qq = q0 + row, key0 = tile * 64 + kb * 32 + c4 * 8;
            if (qq < p.q_len && key0 < p.N) {
              T* dst = p.scores + b * p.sc_sb + h * p.sc_sh + (int64_t)qq * p.sc_sq + key0;
              if (key0 + 8 <= p.N) *reinterpret_cast<u32x4*>(dst) = piece;
              else {
                const T* pe = reinterpret_cast<const T*>(&piece);
                for (int e = 0; e < p.N - key0; ++e) dst[e] = pe[e];
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      } else if (qvalid) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = tile * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key < p.N) stashrow[key] = DT<T>::from_f32(s[kb][r]);
          }
      }
    }
    if (COLIMP) {
      // column sums over this wave's 32 queries (reference-mode importance, kv_cache_token_pruning.py:51, acausal
      // logits included) on the matrix pipe: the rounded logits are exact model-dtype values, so S·Perm with a 0/1
      // permutation matrix is an exact TRANSPOSE into the accumulator layout — afterwards a lane owns one KEY and 16
      // of the 32 queries, and the sum over queries is 15 adds + one lane <-> lane+32 exchange, then ONE atomic
      // instruction for the 32 keys (instead of 5 cross-lane adds per score and an atomic per register).
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) ct[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          frag sf;
#pragma unroll
          for (int e = 0; e < 8; ++e) sf[e] = DT<T>::from_f32(qvalid ? s[kb][t * 8 + e] : 0.f);
          ct = Mfma<T>::mma(sf, perm[t], ct);
        }
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cs += ct[r];
        cs = xor32_sum(cs);
        const int ckey = tile * 64 + kb * 32 + qi;
        if (hi == 0 && ckey < p.N) atomicAdd(colrow + ckey, cs);
      }
    }
    if (!attend) return;                            // a tile above this wave's diagonal: scored for the by-products only
    float m_new, m_base;                            // new running max; the max the exponentials are taken against
    if (!MASK && !edge) {
      // (3a) fully visible tile.  Deferred rescale: the running maximum only moves (and O is only rescaled: 64
      //      multiplies per lane) when some row of the wave outgrew it by more than kDeferMax, see (3c)
      float mt[2] = {-INFINITY, -INFINITY};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt[kb] = max3_raw(mt[kb], s[kb][r], s[kb][r + 1]);
      const float m_tile = xor32_max(fmaxf(mt[0], mt[1]));
      const bool move = __builtin_amdgcn_ballot_w64(m_tile - m_run > kDeferMax) != 0;
      m_new = move ? fmaxf(m_run, m_tile) : m_run;
      m_base = m_new;
    } else {
      // (3b) explicit mask and / or a tile that straddles the causal diagonal: per-element visibility
      float m_tile = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = s[kb][r];
          const int key = tile * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (MASK) { if (key < p.N) v = DT<T>::round(v + DT<T>::to_f32(maskrow[key])); }   // :132
          v = (key < my_vis) ? v : -INFINITY;
          s[kb][r] = v;
          m_tile = fmaxf(m_tile, v);
        }
      m_tile = xor32_max(m_tile);
      m_new = fmaxf(m_run, m_tile);
      m_base = (m_new == -INFINITY) ? 0.f : m_new;  // a fully masked row: exp2(-inf) = 0 for every key
    }
    const float m2 = m_base * kLog2e;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};             // independent partial sums: no 32-deep dependent add chain
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // exp(s - m) as one fma + v_exp_f32 (a base-2 exponential)
          const float pvv = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e], kLog2e, -m2));
          ls[e & 3] += pvv;
          pf[kb][t][e] = DT<T>::from_f32(pvv);
        }
    if (m_new != m_run) {
      const float alpha = __expf(m_run - m_base);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m_run = m_new;
    }
    l_run += (ls[0] + ls[1]) + (ls[2] + ls[3]);
  };

  // ---- prologue (all 8 waves together): K(0) parked in stage 1's K area, stage 0 = { K(1), Vt(0) } --------------
  load_k(0);
  write_k(k_area(1));
  load_stage(0);
  write_stage(0);
  __syncthreads();
  if (wave_tiles > 0) qk(k_area(1));
  __syncthreads();                                   // everyone is done with K(0): stage 1 may be filled
  int next_stage = 1;                                // the stage whose pieces this thread holds in registers
  if (next_stage < n_tiles) load_stage(next_stage);
  if (grp == 1) {                                    // half 1 publishes its stage-1 pieces now, half 0 in its first vector phase
    if (next_stage < n_tiles) { write_stage(next_stage); ++next_stage; if (next_stage < n_tiles) load_stage(next_stage); }
  }
  if (wave_tiles > 0) softmax_tile(0);
  __syncthreads();
  if (grp == 1) __syncthreads();                     // hold half 1 one phase behind

  for (int t = 0; t < n_tiles; ++t) {
    // ---- matrix phase of iteration t -----------------------------------------------------------------------
    PF_STAMP(0);
    if (t < wave_att_tiles) pv(v_area(t));
    if (t + 1 < wave_tiles) qk(k_area(t));
    PF_STAMP(1);
    __syncthreads();
    PF_STAMP(2);
    // ---- vector phase ----------------------------------------------------------------------------------------
    if (next_stage < n_tiles) {
      write_stage(next_stage);
      ++next_stage;
      if (next_stage < n_tiles) load_stage(next_stage);
    }
    PF_STAMP(3);
    if (!(SPATTEN_PF_EXPMODE & 8) && t + 1 < wave_tiles) softmax_tile(t + 1);
    PF_STAMP(4);
    __syncthreads();
    PF_STAMP(5);
  }
  if (grp == 0) __syncthreads();                     // same barrier count for both halves

  // ---- epilogue: O = O^T / l, 4 consecutive dv per 8-byte store ---------------------------------
  const float l_tot = xor32_sum(l_run);
  const float inv = 1.f / l_tot;
  if (p.lse != nullptr && qvalid && hi == 0) {
    float* ls = p.lse + ((int64_t)(b * p.H + h) * p.q_len + myq) * 2;
    ls[0] = m_run; ls[1] = l_tot;
  }
  if (qvalid) {
    T* orow = p.out + b * p.out_sb + (int64_t)myq * p.out_sq + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = db * 32 + 8 * g + 4 * hi;
        T v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = DT<T>::from_f32(o[db][4 * g + e] * inv);
        *reinterpret_cast<u32x2*>(orow + dv) = *reinterpret_cast<u32x2*>(v4);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// (3c) ping-pong flash kernel, 128-key tiles, K / Vt tiles brought in by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write).  Same phase scheme as (3b); a phase now holds 64 MFMAs / 64 scores per lane, which halves
// the per-phase fixed costs (barrier skew, pipeline fill, first-read latency) per unit of work.
// Stage j = { K(j+1), Vt(j) } in slot j & 1 (64 KiB each at d = 128).  Every wave issues its 1-KiB pieces of stage
// j+2 at the top of global phase 2j+2 — half 0 is then entering its matrix phase, half 1 its vector phase; the slot's
// previous tenant (stage j) was last read in phase 2j+1 — and waits for them (vmcnt) before the barrier that ends
// phase 2j+3; the stage is first read in phase 2j+4.
// A DMA instruction writes 1 KiB of LDS linearly (lane l -> +16 l), so the XOR swizzle of the 16-byte slots is applied
// on the GLOBAL side: lane (row, p) fetches logical slot p ^ f(row) of its row.
// ------------------------------------------------------------------------------------------------
// (the LDS-DMA helper dma16 and the MFMA wrappers: mfma_tiles.h)

template <int ROWB> __device__ inline int swz_slot(int row, int p) {   // logical slot stored at physical slot p
  return ROWB == 256 ? (p ^ (row & 15)) : (p ^ ((row >> 1) & 7));
}

template <typename T, int D, bool MASK, int PQK = 0, bool FASTN = false, bool VTRP = false, bool PAIR = false>
__global__ __launch_bounds__(512, 1) void prefill_pp128_kernel(const FlashParams<T> p) {
  constexpr bool FAST = FASTN || (SPATTEN_PF_EXPMODE & 1);
  // (P1 is neutral in the two-half ping-pong — 8192: 691 vs 679 us — and pays where a half runs ALONE, i.e. in the paired form's
  //  solo steps: q = N = 2048 71.2 -> 67.7 us; lock-step halves instead of the ping-pong: 774 us at 8192, 728 with P1)
  constexpr bool P1 = (SPATTEN_PF_P1 || PAIR) && !FAST && PQK == 0 && !MASK;
  constexpr bool NOMAX = (SPATTEN_PF_DIET & 2) && PQK == 0;      // (pass 1 of the quantised keys tracks the row's TRUE maximum)
  constexpr bool DOTSUM = (SPATTEN_PF_DIET & 4) && DT<T>::k16;
  constexpr int KT = 128, NKB = KT / 32;                      // keys per tile, 32-key blocks per tile
  constexpr int KK = D / 16, DB = D / 32, KROWB = D * 2;
  constexpr int KBYTES = KT * KROWB, VBYTES = D * 256, BUF = KBYTES + VBYTES;   // Vt row = 128 keys = 256 B
  constexpr int KINST = KBYTES / 1024 / 8, VINST = VBYTES / 1024 / 8;           // DMA instructions per wave per tile
  using frag = typename Mfma<T>::frag;
  // PQK: + a 4-deep ring of per-key scale vectors (128 fp32 per key tile; tile T in slot T & 3: written by the DMA
  // that brings K(T) in, read by the softmax of tile T one phase later, overwritten four tiles on)
  constexpr int SCALE_OFF = 2 * BUF;
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF + (PQK ? 4 * KT * 4 : 0)];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, qi = lane & 31, hi = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int grp = wave_u >> 2;
  const int nqb = p.nqb;
  int h, qblk, b;
  const int ksplit = PQK == 1 ? 1 : p.ksplit;                       // (pass 2 of the quantised keys may split its key tiles too)
  // which range of key tiles.  Pass 2 of the quantised keys (PQK == 2) lays its grid out range-major: first the B H nqb
  // workgroups of range 0 — all that stay when the pass decides NOT to split (many flagged rows): spread over the XCDs, where
  // item-major the survivors of an 8-way split all landed on ONE XCD (workgroup i runs on XCD i % 8: 4.0 ms instead of 0.5) —
  // then, for every further range, only the list blocks a split pass can have (a split needs <= q_len / 4 flagged rows).
  const int nq4 = (nqb + 3) / 4;
  int ks = 0, item = (int)blockIdx.x, nq_dec = nqb;
  if (ksplit > 1) {
    if (PQK == 2) {
      const int n_full = p.B * p.H * nqb, n_q = p.B * p.H * nq4;
      if ((int)blockIdx.x >= n_full) { const int r_ = (int)blockIdx.x - n_full; ks = 1 + r_ / n_q; item = r_ % n_q; nq_dec = nq4; }
    } else {
      ks = (int)blockIdx.x % ksplit;
      item = (int)blockIdx.x / ksplit;
    }
  }
  {
    const int i = item;
    const int per_b = p.H * nq_dec;
    b = i / per_b;
    const int j = i - b * per_b;
    if ((p.H & 7) == 0) {
      const int xx = j & 7, ss = j >> 3;
      h = xx + 8 * (ss / nq_dec);
      qblk = nq_dec - 1 - (ss % nq_dec);
    } else {
      h = j / nq_dec;
      qblk = nq_dec - 1 - (j % nq_dec);
    }
    // the slot of this (b, h, block) in the partial buffers: the full-grid item (what prefill_merge_kernel computes)
    if (nq_dec != nqb) item = b * p.H * nqb + (((p.H & 7) == 0) ? (h & 7) + 8 * ((h >> 3) * nqb + (nqb - 1 - qblk)) : h * nqb + (nqb - 1 - qblk));
  }
  const int hkv = p.Hkv == p.H ? h : h / (p.H / p.Hkv);
  // PAIR (round 4; causal, no more workgroups than CUs): a 256-row block of a causal prefill needs 2, 4, ... 2 nqb key tiles, and
  // with B * H * nqb <= 256 workgroups nothing balances that — the launch lasts as long as its last block while the mean is half
  // of it (q = N = 2048: 426-437 TFLOP/s against 800+ at 8192).  Here the two halves of a workgroup take DIFFERENT 128-row blocks
  // of the same head, i and n128 - 1 - i: they walk the same key tiles in lock step (the K / V stream is shared as before), the
  // short block's half simply runs out of tiles and from then on only serves DMA and barriers, and the other half — alone on its
  // SIMDs — runs its remaining tiles without the partner's contention: q = N = 2048 437 -> 481 TFLOP/s (71.5 against 78.6 us
  // with the V transpose), 1024 +9 %; with more workgroups than CUs the dispatch order balances better (8192: -15 % when forced) —
  // the host enables it for batch x heads x q_len / 256 <= 256.  (r04 also built the second act — the finished half stores its
  // rows and joins the other half's block on every second remaining tile, partials folded through LDS — correct and NOT faster
  // (75.6 us): a step is bound by the 64-KB tile fill of its CU, not by the halves' arithmetic.  Removed.)
  constexpr bool paired = PAIR;        // (its own instantiation: the plain kernel keeps its register allocation)
  const int blk128 = paired ? (grp == 0 ? 2 * nqb - 1 - qblk : qblk) : 0;
  // PQK == 2 (the LSB-refetch pass, RequantDecision.scala:44-72 / SpAttenController.scala:402): only the rows pass 1
  // flagged are recomputed, ONCE, from the 8-bit keys.  Round 5: the flagged rows of the head arrive as a compacted ascending
  // list; workgroup `qblk` serves list entries [256 qblk, +256) — lane (wave, qi) owns entry 256 qblk + 32 wave + qi, whatever
  // query row that is — and leaves at once when the list is shorter; a wave whose entries lie past the list's end keeps
  // serving the tile DMA and the barriers but computes nothing.  Rows of a wave ascend: its first row bounds the tiles every
  // lane sees in full, its last row the tiles it needs at all.
  int list_cnt = 0, q0_l = 0, qlast_l = -1, blkend_l = 0, myq_l = 0;
  bool wave_live = true;
  if (PQK == 2) {
    list_cnt = p.row_cnt[b * p.H + h];
    if (qblk * 256 >= list_cnt) return;
    const int32_t* rowl = p.rows + (int64_t)(b * p.H + h) * p.q_len;
    const int wpos = qblk * 256 + wave_u * 32;
    wave_live = wpos < list_cnt;
    q0_l = wave_live ? rowl[wpos] : p.q_len;
    qlast_l = wave_live ? rowl[min(wpos + 31, list_cnt - 1)] : -1;
    blkend_l = rowl[min(qblk * 256 + 255, list_cnt - 1)] + 1;
    myq_l = (wpos + qi) < list_cnt ? rowl[wpos + qi] : p.q_len;
  }
  const int q0 = PQK == 2 ? q0_l : (paired ? blk128 * 128 + (wave & 3) * 32 : qblk * 256 + wave * 32);
  const int myq = PQK == 2 ? myq_l : q0 + qi;
  const bool qvalid = myq < p.q_len;
  const int P = p.N - p.q_len;
  const float rsqrt_d = 1.0f / p.sqrt_d;
  const int my_flag = qvalid ? 1 : 0;        // (every listed row is a flagged row)
  // pass 2 splits its key tiles only when few rows were flagged (the rule is the merge kernel's too): a long list fills the chip
  const int ks_eff = (PQK == 2 && ksplit > 1 && (long long)list_cnt * 4 > p.q_len) ? 1 : ksplit;
  if (PQK == 2 && ks >= ks_eff) return;

  // Q fragments, rotated here (modify_llama.py:92, the reference's three rounded ops): fragment kk holds elements
  // [16kk + 8hi, +8) of the row, so kk and kk + KK/2 are exactly the (x[i], x[i + d/2]) pairs RoPE combines
  frag qf[KK];
  {
    const int qq = min(myq, p.q_len - 1);
    const T* qrow = p.q + b * p.q_sb + h * p.q_sh + (int64_t)qq * p.q_sq;
    int ps = p.pos_ids ? (int)p.pos_ids[b * p.pos_sb + qq] : p.pos_q0 + qq;
    ps = min(max(ps, 0), p.table_rows - 1);
    const T* cr = p.cos + (int64_t)ps * (D / 2);
    const T* sr = p.sin + (int64_t)ps * (D / 2);
#pragma unroll
    for (int kk = 0; kk < KK / 2; ++kk) {
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + 16 * kk + 8 * hi), xlo);
      Vec8<T>::unpack(Vec8<T>::ldg(qrow + D / 2 + 16 * kk + 8 * hi), xhi);
      Vec8<T>::unpack(Vec8<T>::ldg(cr + 16 * kk + 8 * hi), cc);
      Vec8<T>::unpack(Vec8<T>::ldg(sr + 16 * kk + 8 * hi), ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[kk][e] = DT<T>::from_f32(ylo[e]); qf[kk + KK / 2][e] = DT<T>::from_f32(yhi[e]); }
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // Row sums on the MATRIX pipe (SPATTEN_PF_ROWSUM_MFMA): O^T = Vt . P^T gets one more 32-row block whose row 0 is all
  // ones — an A operand that lives in registers — so accumulator register 0 of lanes 0-31 IS sum_keys P for the lane's
  // query.  8 extra MFMAs per 128-key tile (the matrix pipe has the slack) replace 64 VALU adds per lane (the vector
  // phase is the pole), and the sum is taken over the SAME rounded P that multiplies V.
  f32x16 osum;
#pragma unroll
  for (int r = 0; r < 16; ++r) osum[r] = 0.f;
  frag ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = DT<T>::from_f32(qi == 0 ? 1.f : 0.f);
  float m_true = -INFINITY;   // PQK == 1: the row's TRUE running maximum (m_run may lag it: deferred rescale)

  const int wg_q_end = PQK == 2 ? blkend_l : min(p.q_len, paired ? (2 * nqb - qblk) * 128 : qblk * 256 + 256);   // (paired: the end of the LATER block)
  const int att_keys = p.causal ? min(p.N, P + wg_q_end) : p.N;
  const int all_tiles = (att_keys + KT - 1) / KT;
  // this workgroup's tiles [T0, T1) (all of them without a key split); an empty range leaves an empty partial
  const int per_split = (all_tiles + ks_eff - 1) / ks_eff;
  const int T0 = min(ks * per_split, all_tiles), T1 = min(T0 + per_split, all_tiles);
  const int n_att_tiles = T1;
  const int n_tiles = T1;
  const int my_vis = p.causal ? min(p.N, P + myq + 1) : p.N;
  const int wave_full_keys = p.causal ? min(p.N, P + q0 + 1) : p.N;
  const int q_past_wave = PQK == 2 ? qlast_l + 1 : min(q0 + 32, p.q_len);      // one past the wave's last query row
  const int wave_att_tiles = !wave_live ? 0 : (p.causal ? min(n_att_tiles, (max(min(p.N, P + q_past_wave), 0) + KT - 1) / KT) : n_att_tiles);
  const int wave_tiles = wave_att_tiles;

  const T* krb = p.kr + b * p.kv_sb + hkv * p.kv_sh;
  constexpr bool VTR = VTRP && D == 128;
  const T* vtb = p.vt + ((int64_t)(b * p.Hkv + hkv) * D) * p.Npad;
  const T* vrb = p.v + b * p.v_sb + hkv * p.v_sh;
  const T* maskrow = MASK ? p.mask + b * p.mask_sb + (int64_t)min(myq, p.q_len - 1) * p.mask_sq : nullptr;

  auto k_area = [&](int stage) -> char* { return lds + (stage & 1) * BUF; };            // holds K(stage + 1)
  auto v_area = [&](int stage) -> char* { return lds + (stage & 1) * BUF + KBYTES; };   // holds Vt(stage)
  // buffer descriptors (wave-uniform): rows past N read as zeros (out of range), no clamping arithmetic per lane
  const int64_t k_bytes = (int64_t)p.N * D * 2, v_bytes = (int64_t)D * p.Npad * 2;
  const float* ksb = PQK ? p.kscale + b * p.ks_sb + hkv * p.ks_sh : nullptr;
  const int64_t ks_bytes = (int64_t)p.N * 4 < 0x7FFFFFFF ? (int64_t)p.N * 4 : 0x7FFFFFFF;
  // This wave's 1-KiB pieces of a tile.  Lane (row-in-piece lr, physical slot ps) fetches logical slot ps ^ f(row); with
  // f(row) = f(first row of the piece) ^ g(lr) (disjoint bits) the per-lane byte offset is ONE xor away from a value
  // computed once per call, and everything wave-uniform (tile, piece) rides in the scalar offset.
  // (opaque_lane: hoisted out of the tile loop the lane terms get spilled, and every reload's vmcnt(0) then
  //  serialises the DMA instructions.)
  auto dma_k = [&](int tile, char* area) {
    constexpr int LPRW = KROWB / 16;                                         // lanes per K row
    const int ln = opaque_lane(lane);
    const int lr = ln / LPRW, ps = ln % LPRW;
    const int base = lr * KROWB + ((ps ^ (KROWB == 256 ? lr : (lr >> 1))) << 4);
#pragma unroll
    for (int i = 0; i < KINST; ++i) {
      const int piece = wave_u * KINST + i;                                  // 1-KiB piece index within the tile
      const int fp = KROWB == 256 ? ((piece * 4) & 15) : ((piece * 4) & 7);  // swizzle key of the piece's first row
      dma16(krb, k_bytes, area + piece * 1024, base ^ (fp << 4), tile * (KT * D * 2) + piece * 1024);
    }
    if (PQK && wave_u == 0) {     // the tile's 128 scale factors: 2 x (64 lanes x 4 bytes), keys past N read as 0
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ksb), 0, (int)ks_bytes, 0x00020000);
      char* dst = lds + SCALE_OFF + (tile & 3) * (KT * 4);
#pragma unroll
      for (int i = 0; i < KT / 64; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * 256), 4, ln * 4,
                                                 (tile * KT + i * 64) * 4, 0, 0);
    }
  };
  // Vt tile: D rows (dv) of 256 B (128 keys); this wave's rows [wave*D/8, +D/8), 4 rows per piece
  auto dma_v = [&](int tile, char* area) {
    const int ln = opaque_lane(lane);
    const int l4 = ln >> 4, ps = ln & 15;
    if (VTR) {
      // row-major value tile: 128 keys x 256 B; a 1-KiB piece = 4 consecutive rows (keys 4p .. 4p+3).  Physical 16-byte
      // slot ps of row r holds logical slot ps ^ 4 (r & 3): the four rows a transposing read touches per 16-lane group then
      // sit in four different 64-byte quarters of the bank space, and the two groups of a half-wave in disjoint halves of
      // each quarter — a half-wave's 32 eight-byte reads cover the 64 banks exactly once.
      const int base = l4 * 256 + ((ps ^ (4 * l4)) << 4);
      const int64_t vr_bytes = (int64_t)p.N * D * 2;       // rows past N read as zeros (out of range)
#pragma unroll
      for (int i = 0; i < VINST; ++i) {
        const int piece = wave_u * VINST + i;
        dma16(vrb, vr_bytes, area + piece * 1024, base, tile * (KT * 256) + piece * 1024);
      }
      return;
    }
    const int base = l4 * p.Npad * 2 + ((ps ^ l4) << 4);
#pragma unroll
    for (int i = 0; i < VINST; ++i) {
      const int piece = wave_u * VINST + i;
      dma16(vtb, v_bytes, area + piece * 1024, base ^ (((piece * 4) & 15) << 4), tile * (KT * 2) + piece * 4 * p.Npad * 2);
    }
  };
  auto dma_stage = [&](int j) {
    if (j + 1 < n_tiles) dma_k(j + 1, k_area(j));
    if (j < n_att_tiles) dma_v(j, v_area(j));
  };
  [[maybe_unused]] auto dma_stage_k = [&](int j) { if (j + 1 < n_tiles) dma_k(j + 1, k_area(j)); };   // (DMA_MODE 3: the stage's two halves
  [[maybe_unused]] auto dma_stage_v = [&](int j) { if (j < n_att_tiles) dma_v(j, v_area(j)); };       //  go out in different phases)

  f32x16 s[NKB];
  frag pf[NKB][2];
#ifndef SPATTEN_PF_RING
#define SPATTEN_PF_RING 4
#endif
  constexpr int RING = SPATTEN_PF_RING;
  auto qk = [&](const char* kbuf) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    frag a[RING];
    // address = (u ^ slot<<4) + block offset: one v_xor per read, nothing to keep in registers across the loop
    const unsigned ku = (unsigned)(kbuf - lds) + qi * KROWB + (KROWB == 256 ? ((qi & 15) << 4) : (((qi >> 1) & 7) << 4));
    auto kfrag = [&](int i) {   // step i: kb = i % NKB, kk = i / NKB
      return *reinterpret_cast<const frag*>(lds + ((ku ^ ((2 * (i / NKB) + hi) << 4)) + (i % NKB) * 32 * KROWB));
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = kfrag(i);
#pragma unroll
    for (int i = 0; i < NKB * KK; ++i) {
      s[i % NKB] = Mfma<T>::mma(a[i % RING], qf[i / NKB], s[i % NKB]);
      if (i + RING < NKB * KK) a[i % RING] = kfrag(i + RING);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, RING, 0);
#pragma unroll
    for (int i = 0; i < NKB * KK - RING; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, RING, 0);
  };
  // P1 (round 4): the two reference roundings of the NEXT tile's logits — 256 of the ~530 VALU instructions of a tile's softmax,
  // and the only part of it that depends on nothing but S(t+1) — ride in the shadow of this wave's OWN P.V MFMAs (S(t+1) is
  // computed first, then P(t).V(t) with eight rounding instructions behind each MFMA) instead of in the vector phase, where
  // they compete with the partner wave's MFMAs for the SIMD's issue port (stripped builds, r04: the flash kernel without its
  // softmax runs 1,454 TFLOP/s against 786 — the vector phase's VALU and the partner's matrix phase ADD rather than overlap).
  auto round_pair = [&](int kb, int r) {
    const f32x2 x = round2<T>(f32x2{s[kb][r], s[kb][r + 1]});
    const f32x2 v = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
    s[kb][r] = v[0];
    s[kb][r + 1] = v[1];
  };
  auto pv = [&](const char* vbuf, auto with_p1) {
    constexpr bool WP1 = decltype(with_p1)::value;
    frag a[RING];
    const unsigned vu = (unsigned)(vbuf - lds) + qi * 256 + ((qi & 15) << 4);
    // VTR: lane (qi, hi) needs, for d = 32 db + qi, the 8 keys its P fragment holds — elements 0..3: keys k16 + 4 hi + 0..3,
    // elements 4..7: keys k16 + 8 + 4 hi + 0..3 with k16 = 32 kb + 16 t (the 32x32 accumulator's row order, the order the Vt
    // pre-pass bakes into its copy) — i.e. two 4-key COLUMN pieces of the row-major tile.  ds_read_b64_tr_b16 hands lane l
    // of a 16-lane group element (row j, column l) of the 4 x 16 block whose row (i >> 2), columns 4 (i & 3) .. +3 lane i
    // points at (measured: tools/mb/tr_read.hip): two reads, 8 rows (2 KiB) apart.
    const int li = lane & 15, kr = li >> 2;
    const unsigned vtu = (unsigned)(vbuf - lds) + (4 * hi + kr) * 256 + ((lane >> 4) & 1) * 32 + ((li & 3) >> 1) * 16 + (li & 1) * 8;
    auto vfrag = [&](int i) {   // step i: db = i % DB, (kb, t) = i / DB
      const int db = i % DB, kt = i / DB, kb = kt >> 1, t = kt & 1;
      if constexpr (VTR) {
        typedef short v4s __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const unsigned a0 = vtu + (kb * 32 + t * 16) * 256 + (((unsigned)db ^ (unsigned)kr) << 6);
        union { v4s h[2]; frag f; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + a0));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + a0 + 2048));
        return u.f;
      } else {
        return *reinterpret_cast<const frag*>(lds + ((vu ^ ((kb * 4 + t * 2 + hi) << 4)) + db * 32 * 256));
      }
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) a[i] = vfrag(i);
#pragma unroll
    for (int i = 0; i < 2 * NKB * DB; ++i) {
      const int db = i % DB, kt = i / DB;
      o[db] = Mfma<T>::mma(a[i % RING], pf[kt >> 1][kt & 1], o[db]);
      if (i + RING < 2 * NKB * DB) a[i % RING] = vfrag(i + RING);
#if SPATTEN_PF_ROWSUM_MFMA
      if (db == DB - 1) osum = Mfma<T>::mma(ones, pf[kt >> 1][kt & 1], osum);
#endif
      if constexpr (WP1) {      // 2 * NKB * DB MFMAs, NKB * 8 logit pairs: pairs_per logit pairs behind each MFMA
        constexpr int pairs_per = (NKB * 8) / (2 * NKB * DB) > 0 ? (NKB * 8) / (2 * NKB * DB) : 1;
#pragma unroll
        for (int j = 0; j < pairs_per; ++j) {
          const int pi = i * pairs_per + j;
          if (pi < NKB * 8) round_pair(pi >> 3, (pi & 7) * 2);
        }
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, RING, 0);
#pragma unroll
    for (int i = 0; i < 2 * NKB * DB - RING; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if constexpr (WP1) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    }
    if constexpr (WP1) {
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, RING + (SPATTEN_PF_ROWSUM_MFMA ? 2 * NKB : 0), 0);
    }
  };

  auto softmax_tile = [&](int tile, bool rounded = false) {
    const bool edge = tile * KT + KT > wave_full_keys;
    if (PQK) {
      // quantised keys: s holds q . q8 (exact integers times the query); the logit is that times the key's scale / sqrt(d)
      // in fp32 — no 16-bit rounding of logits here (the reference has no quantised path; oracle: pq_prefill_attention).
      // Register r of block kb is key (r & 3) + 8 (r >> 2) + 4 hi: four consecutive scales per ds_read_b128
      const float* sl = reinterpret_cast<const float*>(lds + SCALE_OFF + (tile & 3) * (KT * 4)) + 4 * hi;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sl + kb * 32 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kb][4 * g + e] *= sv[e];
        }
    } else {
    // both reference roundings of every logit (matmul -> dtype, "/ sqrt(d)" -> dtype, modify_llama.py:111-113), two
    // scores at a time: at large logits a 16-bit ulp is a visible change of P
    if (!(FAST && !MASK && !edge) && !rounded) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if constexpr (FAST) {
          s[kb][r] *= rsqrt_d; s[kb][r + 1] *= rsqrt_d;
        } else {
          const f32x2 x = round2<T>(f32x2{s[kb][r], s[kb][r + 1]});
          const f32x2 v = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
          s[kb][r] = v[0];
          s[kb][r + 1] = v[1];
        }
      }
    }
    }
    const float sc2 = (FAST && !PQK && !MASK && !edge) ? kLog2e * rsqrt_d : kLog2e;
    // P = exp(s - m_base) as one fma + v_exp_f32 (a base-2 exponential) per logit, packed into the P.V operand; returns this
    // lane's sum of them (its 64 of the row's 128 keys)
    auto exp_pass = [&](float m_base) -> float {
      const float m2 = m_base * kLog2e;
      float ls[4] = {0.f, 0.f, 0.f, 0.f};             // independent partial sums
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
#if SPATTEN_PF_EXPMODE & 16      // anatomy: what do the 64 transcendentals per wave-tile cost?  (WRONG results)
            const float pvv = fmaf(s[kb][t * 8 + e], sc2, -m2);
#else
            const float pvv = __builtin_amdgcn_exp2f(fmaf(s[kb][t * 8 + e], sc2, -m2));
#endif
#if !SPATTEN_PF_ROWSUM_MFMA
            if (!DOTSUM) ls[e & 3] += pvv;
#endif
            pf[kb][t][e] = DT<T>::from_f32(pvv);
          }
#if !SPATTEN_PF_ROWSUM_MFMA
          if constexpr (DOTSUM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] = pair_sum<T>(reinterpret_cast<const uint32_t*>(&pf[kb][t])[j], ls[j]);
          }
#endif
        }
      return (ls[0] + ls[1]) + (ls[2] + ls[3]);
    };
    auto tile_max = [&]() -> float {
      float mt[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        mt[kb] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt[kb] = max3_raw(mt[kb], s[kb][r], s[kb][r + 1]);
      }
      return xor32_max(fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3]))) * (FAST && !PQK ? rsqrt_d : 1.0f);
    };
    float m_new, lsum;                              // new running max (= the max the exponentials are taken against)
    if (!MASK && !edge) {
      // fully visible tile.  Deferred rescale: the running maximum only moves (and O is only rescaled: 64 multiplies
      // per lane) when some row of the wave outgrew it by more than kDeferMax; until then P = exp(s - m_run) <=
      // e^kDeferMax, which bf16 P (constant relative precision) and the fp32 sums carry without loss.  O / l is
      // mathematically unchanged.  All of the previous tile's P·V is already in O (matrix phase, program order), so
      // the decision covers it.
      if constexpr (NOMAX) {
        // r05: no maximum at all on the common path (32 v_max3 + the exchange per tile).  The exponentials are taken against
        // the running maximum as it stands; the lane's sum of its 64 bounds every one of them, so "sum <= kSumBound" proves
        // that none outgrew e^kDeferMax — the deferred rule's own condition.  Otherwise (first tile: m_run = -inf gives
        // inf; an outlier key; NaN) the tile is redone against its true maximum: P and the sum are overwritten, nothing of
        // the first pass survives.
        m_new = m_run;
        lsum = exp_pass(m_run);
        if (__builtin_amdgcn_ballot_w64(!(lsum <= kSumBound)) != 0) {
          m_new = fmaxf(m_run, tile_max());
          lsum = exp_pass(m_new);
        }
      } else {
        const float m_tile = tile_max();
        const bool move = __builtin_amdgcn_ballot_w64(m_tile - m_run > kDeferMax) != 0;   // -inf start: inf > thr
        m_new = move ? fmaxf(m_run, m_tile) : m_run;
        if (PQK == 1) m_true = fmaxf(m_true, m_tile);
        lsum = exp_pass(m_new);
      }
    } else {
      // explicit mask and / or a tile that straddles the causal diagonal: per-element visibility
      float m_tile = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = s[kb][r];
          const int key = tile * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (MASK) { if (key < p.N) v = DT<T>::round(v + DT<T>::to_f32(maskrow[key])); }   // :132
          v = (key < my_vis) ? v : -INFINITY;
          s[kb][r] = v;
          m_tile = fmaxf(m_tile, v);
        }
      m_tile = xor32_max(m_tile);
      if (PQK == 1) m_true = fmaxf(m_true, m_tile);
      m_new = fmaxf(m_run, m_tile);
      lsum = exp_pass((m_new == -INFINITY) ? 0.f : m_new);   // a fully masked row: exp2(-inf) = 0 for every key
    }
    if (m_new != m_run) {
      const float alpha = __expf(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
      l_run *= alpha;
      osum[0] *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m_run = m_new;
    }
    l_run += lsum;
  };

  // ---- prologue (all 8 waves together): K(0) parked in stage 1's K area, stage 0 = { K(1), Vt(0) } --------------
  dma_k(T0, k_area(T0 + 1));
  dma_stage(T0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14)
  __syncthreads();
  if (wave_tiles > T0) qk(k_area(T0 + 1));
  __syncthreads();                                   // everyone is done with K(T0): stage T0+1 may be filled
  if (T0 + 1 < n_tiles) dma_stage(T0 + 1);
  if (wave_tiles > T0) softmax_tile(T0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (!SPATTEN_PF_LOCKSTEP && grp == 1) __syncthreads();                     // hold half 1 one phase behind

#if SPATTEN_PF_PRIO == 1
  if (grp == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the later-dispatched half (A/B knob)
#elif SPATTEN_PF_PRIO == 2
  if (grp == 0) __builtin_amdgcn_s_setprio(1);
#endif
  for (int t = T0; t < n_tiles; ++t) {
    // ---- matrix phase of iteration t -----------------------------------------------------------------------
    PF_STAMP(0);
#if SPATTEN_PF_DMA_MODE == 1
    // BOTH halves bring their pieces of stage t+1 in at the top of their own matrix phase, where the ~115 cycles each DMA
    // instruction takes to issue hide under the MFMAs (phase stamps, r02: issued from the vector phase, half 1's 16 DMA
    // instructions cost 1.8k cycles on top of its 3.5k-cycle softmax, and half 0 idled 1.8k cycles at the barrier every
    // tile).  Half 0 issues in global phase 2t and has the whole following vector phase for them to land; half 1 issues
    // in phase 2t+1 and waits for them at the end of that same phase (stage t+1 is first read in phase 2t+2).  The
    // slot's previous tenant, stage t-1, was last read in phase 2t-1.
    if (!(SPATTEN_PF_EXPMODE & 2) && t >= T0 + 1 && t + 1 < n_tiles) dma_stage(t + 1);
    PF_STAMP(6);           // (the stage's LDS-DMA pieces issued)
#elif SPATTEN_PF_DMA_MODE == 0
    if (grp == 0 && t >= T0 + 1 && t + 1 < n_tiles) dma_stage(t + 1);
#elif SPATTEN_PF_DMA_MODE == 3
    // round 6: the stage's K pieces from the matrix phase (as mode 1), its V pieces from the vector phase (as mode 2): four DMA
    // instructions per wave in each phase instead of eight in one — each costs ~115-125 cycles of the wave's issue whichever phase
    // it sits in, and the phase that carries all eight is the longer one of the step (matrix phase in fast numerics)
    if (!(SPATTEN_PF_EXPMODE & 2) && t >= T0 + 1 && t + 1 < n_tiles) dma_stage_k(t + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    bool rounded = false;
    if constexpr (P1) {
      // S(t+1) first, then P(t).V(t) with the roundings of S(t+1) in its shadow
      const bool nxt = t + 1 < wave_tiles;
      if (nxt) qk(k_area(t));
      __builtin_amdgcn_sched_barrier(0);
      if (t < wave_att_tiles) {
        if (nxt) { pv(v_area(t), std::true_type{}); rounded = true; }
        else pv(v_area(t), std::false_type{});
      }
    } else {
      if (t < wave_att_tiles) pv(v_area(t), std::false_type{});
      __builtin_amdgcn_sched_barrier(0);                   // P is dead from here on: keep S(t+1) out of its live range
      PF_STAMP(7);         // (the P.V MFMAs issued)
      if (t + 1 < wave_tiles) qk(k_area(t));
    }
    PF_STAMP(1);

    if (!SPATTEN_PF_LOCKSTEP && grp == 1) __builtin_amdgcn_s_waitcnt(0x0F70);    // half 1's pieces (issued one phase ago) have landed
    __syncthreads();
    PF_STAMP(2);
    // ---- vector phase ----------------------------------------------------------------------------------------
#if SPATTEN_PF_DMA_MODE == 0
    if (grp == 1 && t + 2 < n_tiles) dma_stage(t + 2);
#elif SPATTEN_PF_DMA_MODE == 2
    // both halves issue from their VECTOR phase (stage t+1 may be written during global phases 2t and 2t+1: half 1 is in
    // its vector phase of iteration t-1 during 2t, half 0 in its vector phase of iteration t during 2t+1)
    if (grp == 0 && t >= T0 + 1 && t + 1 < n_tiles) dma_stage(t + 1);
    if (grp == 1 && t + 2 < n_tiles) dma_stage(t + 2);
#elif SPATTEN_PF_DMA_MODE == 3
    if (grp == 0 && t >= T0 + 1 && t + 1 < n_tiles) dma_stage_v(t + 1);
    if (grp == 1 && t + 2 < n_tiles) dma_stage_v(t + 2);
#endif
    PF_STAMP(3);
    if (!(SPATTEN_PF_EXPMODE & 8) && t + 1 < wave_tiles) softmax_tile(t + 1, rounded);
    PF_STAMP(4);
    if (SPATTEN_PF_LOCKSTEP || grp == 0) __builtin_amdgcn_s_waitcnt(0x0F70);    // half 0's pieces (issued at the top of this iteration)
    __syncthreads();
    PF_STAMP(5);
  }
  if (!SPATTEN_PF_LOCKSTEP && grp == 0) __syncthreads();

  // ---- epilogue: O = O^T / l, 4 consecutive dv per 8-byte store ---------------------------------
#if SPATTEN_PF_ROWSUM_MFMA
  const float l_tot = xor32_sum(hi == 0 ? osum[0] : 0.f);     // register 0 of lanes 0-31 = row 0 of the ones block
#else
  const float l_tot = xor32_sum(l_run);
#endif
  if (ks_eff > 1) {       // partial result of this key range: un-normalised O^T, (m, l) — folded by prefill_merge_kernel
    const int64_t slot = ((int64_t)item * ksplit + ks) * 256 + wave * 32 + qi;
    float* po = p.part_o + slot * D;
    if (qvalid) {           // rows past q_len are never read back
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = db * 32 + 8 * g + 4 * hi;
          *reinterpret_cast<f32x4*>(po + dv) = f32x4{o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
        }
      if (hi == 0) { p.part_ml[slot * 2] = m_run; p.part_ml[slot * 2 + 1] = l_tot; }
    }
    return;
  }
  const float inv = 1.f / l_tot;
  if (p.lse != nullptr && qvalid && hi == 0) {
    float* ls = p.lse + ((int64_t)(b * p.H + h) * p.q_len + myq) * 2;
    ls[0] = m_run; ls[1] = l_tot;
  }
  if (PQK == 1) {   // need_lsb = max_j prob_j < threshold (RequantDecision.scala:44-72): exp(max - reference max) / sum
    const float pmax = __expf(m_true - m_run) * inv;
    if (qvalid && hi == 0) p.need[(int64_t)(b * p.H + h) * p.q_len + myq] = pmax < p.pq_thr ? 1 : 0;
  }
#ifndef SPATTEN_PF_ROWSTORE    // 1: O staged through LDS and stored as whole rows (A/B: tools/mb/pf_exp.sh SPATTEN_PF_ROWSTORE 0 1)
#define SPATTEN_PF_ROWSTORE 1
#endif
  if (SPATTEN_PF_ROWSTORE && PQK != 2 && D == 128) {
    // Whole rows: the accumulator layout gives a lane 4 consecutive dv of ONE query row per store — 16 eight-byte stores per
    // lane at a row stride, 32-64 lines touched by every wave instruction; the epilogue of a 256-row block then costs ~9k
    // cycles of store issue (a tile is ~8k: 13 % of a q = N = 2048 block).  All tile traffic is over (the barrier above), so
    // each wave transposes its 32 rows through its own 8.5 KB of the tile buffers (pitch 272 B: the 32 row starts spread over
    // the banks) and stores 4 rows of 256 contiguous bytes per instruction.
    constexpr int PITCH = D * 2 + 16;
    char* stg = lds + wave * (32 * PITCH);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = db * 32 + 8 * g + 4 * hi;
        T v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = DT<T>::from_f32(o[db][4 * g + e] * inv);
        *reinterpret_cast<u32x2*>(stg + qi * PITCH + dv * 2) = *reinterpret_cast<u32x2*>(v4);
      }
    const int chunk = lane & 15, r4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + r4;
      const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * PITCH + chunk * 16);
      if (q0 + row < p.q_len)
        *reinterpret_cast<u32x4*>(p.out + b * p.out_sb + (int64_t)(q0 + row) * p.out_sq + h * D + chunk * 8) = v;
    }
    return;
  }
  if (qvalid && my_flag) {
    T* orow = p.out + b * p.out_sb + (int64_t)myq * p.out_sq + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = db * 32 + 8 * g + 4 * hi;
        T v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = DT<T>::from_f32(o[db][4 * g + e] * inv);
        *reinterpret_cast<u32x2*>(orow + dv) = *reinterpret_cast<u32x2*>(v4);
      }
    }
  }
}


// (The one-wave-per-SIMD flash kernel of round 3 — measured slower, 616-642 against 740-766 TFLOP/s — was removed in round 5:
//  HISTORY.md, `git show 8a7875c:tools/experiments/prefill_w4.h`.)

// Fold the key-split partials of prefill_pp128_kernel: one wave per query row (D/64 output elements per lane).
// out = sum_s O_s e^(m_s - m) / sum_s l_s e^(m_s - m),  m = max_s m_s   (a split without keys has m = -inf, l = 0)
constexpr int kMergeListWgs = 32;
template <typename T, int D>
__device__ __forceinline__ void prefill_merge_row(const FlashParams<T>& p, int bh, int lpos, int qi, int lane);
template <typename T, int D>
__global__ __launch_bounds__(256) void prefill_merge_kernel(const FlashParams<T> p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (p.rows != nullptr) {
    // the refetch pass of the quantised keys: rows are LIST entries of a (b, h); kMergeListWgs workgroups per (b, h) walk the
    // list (the count lives on the device: a grid over every possible row was 65,536 workgroups that mostly left at once);
    // nothing to fold when the pass did not split (its own rule)
    const int bh = (int)blockIdx.x / kMergeListWgs, w0 = ((int)blockIdx.x % kMergeListWgs) * 4 + wave;
    const int cnt = p.row_cnt[bh];
    if ((long long)cnt * 4 > p.q_len) return;
    for (int lpos = w0; lpos < cnt; lpos += kMergeListWgs * 4) prefill_merge_row<T, D>(p, bh, lpos, p.rows[(int64_t)bh * p.q_len + lpos], lane);
    return;
  }
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;                 // over B * H * q_len
  if (row >= (int64_t)p.B * p.H * p.q_len) return;
  prefill_merge_row<T, D>(p, (int)(row / p.q_len), (int)(row % p.q_len), (int)(row % p.q_len), lane);
}

template <typename T, int D>
__device__ __forceinline__ void prefill_merge_row(const FlashParams<T>& p, int bh, int lpos, int qi, int lane) {
  const int h = bh % p.H, b = bh / p.H;
  // lpos = position inside the (b, h)'s blocks: the row itself, or its list entry; qi = the query row written
  // the flash kernel's work order: item = b * H * nqb + j with j as decoded there
  const int qblk = lpos / 256, qin = lpos % 256, nqb = p.nqb;
  int j;
  if ((p.H & 7) == 0) j = (h & 7) + 8 * ((h >> 3) * nqb + (nqb - 1 - qblk));
  else j = h * nqb + (nqb - 1 - qblk);
  const int64_t item = (int64_t)b * p.H * nqb + j;
  const int S = p.ksplit;
  float m = -INFINITY;
  for (int s = 0; s < S; ++s) m = fmaxf(m, p.part_ml[((item * S + s) * 256 + qin) * 2]);
  const float mu = (m == -INFINITY) ? 0.f : m;
  float l = 0.f, acc[D / 64];
#pragma unroll
  for (int e = 0; e < D / 64; ++e) acc[e] = 0.f;
  for (int s = 0; s < S; ++s) {
    const int64_t slot = (item * S + s) * 256 + qin;
    const float w = __expf(p.part_ml[slot * 2] - mu);               // exp(-inf) = 0
    l = fmaf(p.part_ml[slot * 2 + 1], w, l);
#pragma unroll
    for (int e = 0; e < D / 64; ++e) acc[e] = fmaf(p.part_o[slot * D + lane + 64 * e], w, acc[e]);
  }
  const float inv = 1.f / l;
  T* orow = p.out + b * p.out_sb + (int64_t)qi * p.out_sq + h * D;
#pragma unroll
  for (int e = 0; e < D / 64; ++e) orow[lane + 64 * e] = DT<T>::from_f32(acc[e] * inv);
  if (p.lse != nullptr && lane == 0) { float* ls = p.lse + ((int64_t)bh * p.q_len + qi) * 2; ls[0] = m; ls[1] = l; }
}

// The flagged query rows of every (b, h) after pass 1 of the quantised-key prefill, compacted in ascending order (the list pass 2
// walks): one workgroup per (b, h), 256 flags per step, positions from ballots + a running count.
__global__ __launch_bounds__(256) void pq_rows_compact_kernel(const int32_t* __restrict__ need, int32_t* __restrict__ rows,
                                                              int32_t* __restrict__ cnt, int q_len) {
  __shared__ int s_w[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t base = (int64_t)blockIdx.x * q_len;
  int run = 0;
  for (int i0 = 0; i0 < q_len; i0 += 256) {
    const int i = i0 + tid;
    const bool f = i < q_len && need[base + i] != 0;
    const unsigned long long m = __ballot(f);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) before += c; total += c; }
    if (f) rows[base + run + before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = i;
    run += total;
    __syncthreads();
  }
  if (tid == 0) cnt[blockIdx.x] = run;
}

// ------------------------------------------------------------------------------------------------
// Cascade importance of a multi-token forward WITHOUT the [B,H,q,N] stash:
//   acc[h, j] += sum over b and the query rows i that see key j of  exp(s_ij - m_i) / l_i
// with s_ij recomputed on the matrix cores (same operands, same two roundings as the flash kernel) and (m_i, l_i) the
// row statistics the flash kernel wrote.  Operand roles are swapped relative to the flash kernel — S = Q K^T with the
// keys as the B operand — so a lane owns ONE KEY (its 32-key block's fragments stay in registers for the whole
// launch) and its accumulator registers enumerate 16 queries: the sum over queries is 16 in-lane adds per block and
// one lane <-> lane+32 exchange at the very end; no atomics inside the loop, no cross-lane traffic per score.
// Workgroup = 256 keys (8 waves x 32), looping over 128-query tiles staged through LDS (XOR-swizzled 16-byte slots).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct ColProbParams {
  const T* qrot; int64_t q_sb, q_sh;      // rotated queries [B,H,q_len,D], rows contiguous
  const T* kr; int64_t kv_sb, kv_sh;      // rotated shadow
  const float* lse;                       // [B,H,q_len,2]
  float* acc; int64_t acc_sh;             // [H, >=N]
  int H, Hkv, q_len, N, causal;
  float sqrt_d;
};

template <typename T, int D>
__global__ __launch_bounds__(512, 1) void prefill_colprob_kernel(const ColProbParams<T> p) {
  constexpr int KK = D / 16, QT = 128, ROWB = D * 2;
  constexpr int QBYTES = QT * ROWB, PIECES = QT * (D / 8) / 512;     // 16-byte pieces per thread per query tile
  using frag = typename Mfma<T>::frag;
  __shared__ __attribute__((aligned(16))) char lds[2 * QBYTES + 2 * QT * 8];   // (2 x QT x 4 B of statistics are used)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ki = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int hkv = p.Hkv == p.H ? h : h / (p.H / p.Hkv);
  const int kblk = (int)blockIdx.x * 256;          // (int: blockIdx is unsigned, and kblk - P below may be negative)
  const int key = kblk + wave * 32 + ki;
  const int P = p.N - p.q_len;
  const float rsqrt_d = 1.0f / p.sqrt_d;
  // this lane's key as the B operand: fragment kk = elements [16kk + 8hi, +8) of the key row (zeros past N)
  frag kf[KK];
  {
    const T* krow = p.kr + b * p.kv_sb + hkv * p.kv_sh + (int64_t)min(key, p.N - 1) * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(krow + 16 * kk + 8 * hi);
      kf[kk] = *reinterpret_cast<const frag*>(&raw);
    }
  }
  const T* qb = p.qrot + b * p.q_sb + h * p.q_sh;
  const float* lb = p.lse + (int64_t)(b * p.H + h) * p.q_len * 2;
  // query tiles that can see any key of this workgroup (causal: row i sees keys j <= P + i)
  const int first_q = p.causal ? max(0, kblk - P) : 0;
  const int t_lo = first_q / QT, t_hi = (p.q_len + QT - 1) / QT;
  u32x4 qreg[PIECES];
  f32x2 sreg;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int id = tid + 512 * i, row = id / (D / 8), slot = id % (D / 8);
      const int qi = min(t * QT + row, p.q_len - 1);
      qreg[i] = *reinterpret_cast<const u32x4*>(qb + (int64_t)qi * D + slot * 8);
    }
    if (tid < QT) {
      const int qi = min(t * QT + tid, p.q_len - 1);
      sreg = *reinterpret_cast<const f32x2*>(lb + (int64_t)qi * 2);
    }
  };
  auto write_tile = [&](int buf) {
    char* base = lds + buf * QBYTES;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int id = tid + 512 * i, row = id / (D / 8), slot = id % (D / 8);
      *reinterpret_cast<u32x4*>(base + lds_off<ROWB>(row, slot)) = qreg[i];
    }
    if (tid < QT) {   // per query of the tile: m * log2e + log2(l), so that exp2(s * log2e - it) IS the probability
      float* st = reinterpret_cast<float*>(lds + 2 * QBYTES + buf * QT * 8);
      st[tid] = sreg[1] > 0.f ? fmaf(sreg[0], kLog2e, __log2f(sreg[1])) : INFINITY;   // a row without keys: probability 0
    }
  };
  float colsum = 0.f;
  if (t_lo < t_hi) {
    load_tile(t_lo);
    write_tile(0);
  }
  __syncthreads();
  for (int t = t_lo; t < t_hi; ++t) {
    const int buf = (t - t_lo) & 1;
    if (t + 1 < t_hi) load_tile(t + 1);
    const char* qt = lds + buf * QBYTES;
    const float* st = reinterpret_cast<const float*>(lds + 2 * QBYTES + buf * QT * 8);
#pragma unroll
    for (int blk = 0; blk < QT / 32; ++blk) {
      const int q0 = t * QT + blk * 32;
      if (q0 >= p.q_len) break;
      if (p.causal && P + q0 + 31 < kblk + wave * 32) continue;      // none of these rows sees this wave's keys
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const frag a = *reinterpret_cast<const frag*>(qt + lds_off<ROWB>(blk * 32 + ki, 2 * kk + hi));
        c = Mfma<T>::mma(a, kf[kk], c);
      }
      // register r <-> query q0 + (r & 3) + 8 (r >> 2) + 4 hi; the two reference roundings of the logit, then its
      // probability.  Blocks every row of which sees every key of this wave (all but the ones on the diagonal and at
      // the ragged ends) skip the per-element visibility test: this kernel is VALU-bound (8 MFMAs per 16 scores a lane)
      const bool all_vis = q0 + 31 < p.q_len && kblk + wave * 32 + 31 < p.N && (!p.causal || kblk + wave * 32 + 31 <= P + q0);
      float pr[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(st + blk * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 x = round2<T>(f32x2{c[4 * g + e], c[4 * g + e + 1]});
          const f32x2 v = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
          pr[4 * g + e] = __builtin_amdgcn_exp2f(fmaf(v[0], kLog2e, -m4[e]));
          pr[4 * g + e + 1] = __builtin_amdgcn_exp2f(fmaf(v[1], kLog2e, -m4[e + 1]));
        }
      }
      if (all_vis) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) { a0 += pr[r]; a1 += pr[r + 1]; a2 += pr[r + 2]; a3 += pr[r + 3]; }
        colsum += (a0 + a1) + (a2 + a3);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool vis = qi < p.q_len && key < p.N && (!p.causal || key <= P + qi);
          colsum += vis ? pr[r] : 0.f;
        }
      }
    }
    __syncthreads();                       // everyone is done with the other buffer's previous tenant
    if (t + 1 < t_hi) write_tile(buf ^ 1);
    __syncthreads();
  }
  colsum = xor32_sum(colsum);
  if (hi == 0 && key < p.N && colsum != 0.f) atomicAdd(p.acc + h * p.acc_sh + key, colsum);
}

constexpr int kRowsPerLaunch = 4096;   // query rows per launch of the rows leg
static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
static inline int rows_leg(int dtype, int head_dim, int q_len) {
  return dtype == SPATTEN_F32 || q_len <= 8 || (head_dim != 64 && head_dim != 128);
}
static inline int rows_splits(int units) { int s = 256 / (units > 0 ? units : 1); return s < 1 ? 1 : (s > 64 ? 64 : s); }

// Kernel choice.  No by-products (the prefill of the multi-turn protocol, the bench): 128-key tiles + LDS-DMA.  With a
// stash / column-importance output the softmax phase carries the extra stores and reductions, and the 64-key kernel
// (half the live scores per lane) is the one that stays out of scratch.  SPATTEN_PREFILL_VARIANT=1 forces the 64-key
// kernel everywhere (A/B measurements).
static int prefill_variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SPATTEN_PREFILL_VARIANT"); v = e ? atoi(e) : 0; if (v < 0 || v > 1) v = 0; }
  return v;
}

// Transposing-read form of the 128-key kernel (no Vt pre-pass): when the pre-pass — all kv_len keys, whatever q_len — is
// a large part of the launch, i.e. a short query block on a long cache.
static inline bool use_vtr_for(int head_dim, int q_len, int kv_len) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("SPATTEN_PREFILL_VTR"); env = e ? atoi(e) : 2; }   // 0 off, 1 on, 2 by shape
  if (head_dim != 128 || env == 0) return false;
  if (env == 1) return true;
  // measured (tools/probe_vtr.py, H = 32, us per layer, Vt form | transposing reads): q = 64 on 2112: 36.2 | 28.7; 256 on
  // 2304: 44.3 | 37.5; 512 on 2560: 56.9 | 51.2; 1024 on 2048: 76.8 | 81.1; 1024 on 4096: 132.5 | 149.3; 2048 on 2048:
  // 79.9 | 83.5; 4096 on 4096: 246 | 265 — the pre-pass costs ~9 us per 2k keys, the two-read fragments ~7 % of the flash time
  // (r04, paired blocks: a first prompt of 1024 tokens — q = N = 1024 — 40.3 | 37.7; 2048 on 2048 68.4 | 68.6;
  //  r05, after the softmax diet: q = N = 1024 39.4 | 35.7, 1536 52.7 | 49.4, 2048 66.3 | 64.1, 2560 104.1 | 101.7, 3072 144.0 |
  //  155.0; 1024 on 2048 68.2 | 69.4, 2048 on 4096 148.4 | 148.4 — a whole prompt up to 2560 tokens takes the transposing reads)
  return q_len <= SPATTEN_PF_VTR_MAXQ || (q_len == kv_len && q_len <= 5 * SPATTEN_PF_VTR_MAXQ);
}


// Key split of the plain flash kernel: only when the launch would leave most of the chip idle (few query blocks x heads)
// and every range still has >= 2 key tiles.  The same rule sizes the workspace.
static inline int flash_ksplit(int batch, int heads, int q_len, int kv_len) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("SPATTEN_PREFILL_KSPLIT"); env = e ? atoi(e) : 0; }   // 1 = off, n = forced (A/B)
  const long long items = (long long)batch * heads * ceil_div(q_len, 256);
  const int tiles = ceil_div(kv_len, 128);
  int ks = (items >= 128 || tiles < 8) ? 1 : (int)(256 / items);    // short key ranges: the partials cost more than they buy
  if (env > 0) ks = env;
  if (ks > 8) ks = 8;
  if (ks > tiles / 2) ks = tiles / 2;
  if (ks < 1) ks = 1;
  const int per = ceil_div(tiles, ks);          // tiles per range ...
  return ceil_div(tiles, per);                  // ... and no empty ranges (17 tiles: 8 ranges of 3 would leave two empty)
}
// paired 128-row blocks (prefill_pp128_kernel<..., PAIR>): a causal block of whole 256-row groups whose workgroups all fit the chip at once
static inline bool flash_pair(int batch, int heads, int q_len, int causal) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("SPATTEN_PREFILL_PAIR"); env = e ? atoi(e) : 2; }   // 0 = off, 1 = forced (A/B), 2 = auto
  if (!causal || q_len % 256 != 0 || q_len < 512 || env == 0) return false;
  // (r05: also when the 256-row blocks are exactly TWO per CU — q = N = 4096 at 32 heads 235.6 -> 226.2 us, 4096 on 8192 525 -> 519;
  //  320 workgroups (2560 tokens) 103 -> 122, 640 / 768 / 1024: slower — the dispatch order balances those by itself)
  const long long items = (long long)batch * heads * (q_len / 256);
  return env == 1 || items <= 256 || items == 512;
}
static inline size_t flash_partial_bytes(int batch, int heads, int head_dim, int q_len, int ks) {
  if (ks <= 1) return 0;
  return (size_t)batch * heads * ceil_div(q_len, 256) * ks * 256 * (head_dim + 2) * sizeof(float);
}

// key ranges of the refetch pass of the quantised-key prefill: >= 8 tiles of 128 keys each, at most 8 ranges (the kernel falls
// back to ONE range by itself when more than a quarter of the rows were flagged: a long list fills the chip without splitting)
static inline int pq_pass2_ksplit(int kv_len) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("SPATTEN_PQ_PASS2_KSPLIT"); env = e ? atoi(e) : 0; }     // 1 = off (A/B)
  const int tiles = ceil_div(kv_len, 128);
  int ks = env > 0 ? env : tiles / 8;
  if (ks > 8) ks = 8;
  if (ks < 1) ks = 1;
  const int per = ceil_div(tiles, ks);
  return ceil_div(tiles, per);
}

template <typename T, int D, bool ST, bool CI>
static void launch_flash_m(const FlashParams<T>& p, hipStream_t st) {
  const dim3 grid((unsigned)(p.nqb * p.H * p.B));
  if constexpr (!ST && !CI) {
    if (p.kscale != nullptr) {          // progressive-quant keys: pass 1 (MSB logits + need flags) or pass 2 (flagged rows)
      if (p.pq_thr >= 0.f) {
        if (p.mask) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, true, 1>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 1>), grid, dim3(512), 0, st, p);
      } else {
        const dim3 grid2((unsigned)(p.H * p.B * (p.nqb + (p.ksplit > 1 ? (p.ksplit - 1) * ((p.nqb + 3) / 4) : 0))));   // (see the kernel)
        if (p.mask) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, true, 2>), grid2, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 2>), grid2, dim3(512), 0, st, p);
        if (p.ksplit > 1)
          hipLaunchKernelGGL((prefill_merge_kernel<T, D>), dim3((unsigned)(p.B * p.H * kMergeListWgs)), dim3(256), 0, st, p);
      }
      return;
    }
    if (prefill_variant() == 0) {
      const dim3 gridk((unsigned)(p.nqb * p.H * p.B * (p.ksplit > 1 ? p.ksplit : 1)));
      if (p.mask) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, true>), gridk, dim3(512), 0, st, p);
      else if (p.fast) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 0, true>), gridk, dim3(512), 0, st, p);
      else if (p.pair && p.vtr) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 0, false, true, true>), gridk, dim3(512), 0, st, p);
      else if (p.pair) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 0, false, false, true>), gridk, dim3(512), 0, st, p);
      else if (p.vtr) hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false, 0, false, true>), gridk, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((prefill_pp128_kernel<T, D, false>), gridk, dim3(512), 0, st, p);
      if (p.ksplit > 1) {
        const long long rows = (long long)p.B * p.H * p.q_len;
        hipLaunchKernelGGL((prefill_merge_kernel<T, D>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
      }
      return;
    }
  }
  if (p.mask) hipLaunchKernelGGL((prefill_pp_kernel<T, D, ST, CI, true>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((prefill_pp_kernel<T, D, ST, CI, false>), grid, dim3(512), 0, st, p);
}

template <typename T, int D>
static int launch_flash(const FlashParams<T>& p, hipStream_t st) {
  const bool st_ = p.scores != nullptr, ci = p.col_imp != nullptr;
  if (st_ && ci) launch_flash_m<T, D, true, true>(p, st);
  else if (st_) launch_flash_m<T, D, true, false>(p, st);
  else if (ci) launch_flash_m<T, D, false, true>(p, st);
  else launch_flash_m<T, D, false, false>(p, st);
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_prefill_workspace_bytes(int dtype, int batch, int heads, int kv_heads, int head_dim,
                                                  int q_len, int kv_len) {
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || head_dim <= 0 || q_len <= 0 || kv_len <= 0) return 0;
  if (rows_leg(dtype, head_dim, q_len)) {
    // the maximum over the slices spatten_attn_prefill launches: full slices of rows_per query rows and a shorter tail,
    // whose FEWER units mean MORE splits per unit (round-2 advisor finding: sizing from the first slice alone under-sized
    // the tail's partial region, e.g. fp32 B = 1, H = 8, q_len = 4097)
    const int rows_per = max(1, min(kRowsPerLaunch, 65535 / batch));
    size_t need = 0;
    const int slices[2] = {min(q_len, rows_per), q_len % rows_per};
    for (int nq : slices) {
      if (nq <= 0) continue;
      const size_t units = (size_t)batch * heads * (size_t)nq;
      const int S = rows_splits((int)(units > (1u << 30) ? (1u << 30) : units));
      const size_t b = kDecodeWsHeader + decode_cnt_bytes(units) + (S > 1 ? units * S * (head_dim + 2) * sizeof(unsigned long long) : 0);
      need = b > need ? b : need;
    }
    return 256 + need;
  }
  const size_t es = 2, npad = (size_t)ceil_div(kv_len, 128) * 128;
  return 256 + align256((size_t)batch * kv_heads * head_dim * npad * es)      // the key-contiguous copy of V
         + align256(flash_partial_bytes(batch, heads, head_dim, q_len, flash_ksplit(batch, heads, q_len, kv_len)));   // key-split partials
}

extern "C" int spatten_attn_prefill(int dtype, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq,
                                    const void* kr_cache, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                                    const void* cos, const void* sin, int table_rows, const int64_t* position_ids,
                                    int64_t pos_sb, const void* mask, int64_t mask_sb, int64_t mask_sq, void* out,
                                    int64_t out_sb, int64_t out_sq, void* scores, int64_t sc_sb, int64_t sc_sh,
                                    int64_t sc_sq, float* col_importance, float* lse, void* workspace, int batch, int heads,
                                    int kv_heads, int head_dim, int q_len, int kv_len, int pos_q0, int causal,
                                    void* stream) {
  if (!q || !kr_cache || !v_cache || !cos || !sin || !out || !workspace) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || q_len <= 0 || kv_len < q_len || pos_q0 < 0)
    return SPATTEN_ERR_INVALID;
  if (dtype != SPATTEN_F32 && dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return SPATTEN_ERR_INVALID;
  if (table_rows < kv_len || (!position_ids && pos_q0 + q_len > table_rows)) return SPATTEN_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)(((uintptr_t)workspace + 255) / 256 * 256);

  if (rows_leg(dtype, head_dim, q_len)) {
    if (col_importance) return SPATTEN_ERR_UNSUPPORTED;   // by-product of the flash leg only (use the stash here)
    // one decode workgroup column per query row; query rows go out in slices of kRowsPerLaunch so that grid.z
    // (B * rows) stays inside the launch limit whatever q_len is.  Every row re-streams K/V (through L2): the cost
    // is O(q_len * kv_len) memory traffic — this leg is for exactness (fp32) and short blocks, not for long prefill.
    const int rows_per = max(1, min(kRowsPerLaunch, 65535 / batch));
    for (int i0 = 0; i0 < q_len; i0 += rows_per) {
      const int nq = min(rows_per, q_len - i0);
      const size_t units = (size_t)batch * heads * nq;
      const int S = rows_splits((int)units);
      const size_t cnt = kDecodeWsHeader + decode_cnt_bytes(units);
      // counters / generations start cleared; granule tags too (a workspace reused across shapes may hold stale ones)
      if (S > 1 && hipMemsetAsync(ws, 0, cnt + units * S * (head_dim + 2) * sizeof(unsigned long long), st) != hipSuccess)
        return SPATTEN_ERR_LAUNCH;
      const int es = dtype == SPATTEN_F32 ? 4 : 2;
      DecodeCall c;
      c.dtype = dtype;
      c.q = (const char*)q + (int64_t)i0 * q_sq * es; c.q_sb = q_sb; c.q_sh = q_sh; c.q_sq = q_sq;
      c.kr_cache = const_cast<void*>(kr_cache); c.v_cache = const_cast<void*>(v_cache); c.kv_sb = kv_sb; c.kv_sh = kv_sh;
      c.cos = cos; c.sin = sin; c.table_rows = table_rows;
      c.position_ids = position_ids ? position_ids + i0 : nullptr; c.pos_sb = pos_sb;
      c.mask = mask ? (const char*)mask + (int64_t)i0 * mask_sq * es : nullptr; c.mask_sb = mask_sb; c.mask_sq = mask_sq;
      c.out = (char*)out + (int64_t)i0 * out_sq * es; c.out_sb = out_sb; c.out_sq = out_sq;
      c.scores = scores ? (char*)scores + (int64_t)i0 * sc_sq * es : nullptr; c.sc_sb = sc_sb; c.sc_sh = sc_sh; c.sc_sq = sc_sq;
      c.lse = lse ? lse + (int64_t)i0 * 2 : nullptr; c.lse_q = q_len;
      c.workspace = ws; c.ws_units = units; c.ws_splits = S;
      c.batch = batch; c.heads = heads; c.kv_heads = kv_heads; c.head_dim = head_dim;
      c.kv_len = kv_len; c.pos_q = pos_q0 + i0; c.n_q = nq; c.causal = causal & 1; c.n_splits = S;
      c.vis0 = kv_len - q_len + i0 + 1;          // HF causal rule for row i0 of the whole block
      const int rc = decode_rows(c, st);
      if (rc != SPATTEN_OK) return rc;
    }
    return SPATTEN_OK;
  }

  const int npad = ceil_div(kv_len, 128) * 128;      // Vt rows padded to whole 128-key tiles (zeros beyond kv_len)
  void* vt = ws;
  // (1) the queries are rotated inside the flash kernel (its prologue)
  // (2) key-contiguous V — only for the kernels that still read it: the 64-key kernel (stash / column-importance outputs,
  // SPATTEN_PREFILL_VARIANT=1) and d = 64; the 128-key kernel at d = 128 reads the value rows themselves (SPATTEN_PF_VTR)
  const bool use_vtr = use_vtr_for(head_dim, q_len, kv_len) && !scores && !col_importance && !mask && prefill_variant() == 0 &&
                       !(causal & SPATTEN_PREFILL_FAST_NUMERICS);
  if (!use_vtr) {
    const dim3 grid((unsigned)(npad / 64), (unsigned)kv_heads, (unsigned)batch);
    if (head_dim == 128)
      hipLaunchKernelGGL((vt_kernel<128>), grid, dim3(256), 0, st, (const uint16_t*)v_cache, kv_sb, kv_sh, (uint16_t*)vt, kv_len, npad, kv_heads);
    else
      hipLaunchKernelGGL((vt_kernel<64>), grid, dim3(256), 0, st, (const uint16_t*)v_cache, kv_sb, kv_sh, (uint16_t*)vt, kv_len, npad, kv_heads);
    if (hipGetLastError() != hipSuccess) return SPATTEN_ERR_LAUNCH;
  }
  // (3) flash
#define SPATTEN_FLASH(T, DD)                                                                           \
  {                                                                                                    \
    FlashParams<T> p;                                                                                  \
    p.rows = nullptr; p.row_cnt = nullptr;                                                             \
    p.q = (const T*)q; p.q_sb = q_sb; p.q_sh = q_sh; p.q_sq = q_sq;                                    \
    p.cos = (const T*)cos; p.sin = (const T*)sin; p.table_rows = table_rows;                           \
    p.pos_ids = position_ids; p.pos_sb = pos_sb; p.pos_q0 = pos_q0;                                    \
    p.kr = (const T*)kr_cache; p.kv_sb = kv_sb; p.kv_sh = kv_sh;                                       \
    p.vt = (const T*)vt; p.mask = (const T*)mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;            \
    p.v = (const T*)v_cache; p.v_sb = kv_sb; p.v_sh = kv_sh; p.vtr = use_vtr ? 1 : 0;                  \
    p.out = (T*)out; p.out_sb = out_sb; p.out_sq = out_sq;                                             \
    p.scores = (T*)scores; p.sc_sb = sc_sb; p.sc_sh = sc_sh; p.sc_sq = sc_sq;                          \
    p.col_imp = col_importance; p.lse = lse; p.kscale = nullptr; p.ks_sb = p.ks_sh = 0; p.need = nullptr; p.pq_thr = 0.f; \
    p.B = batch; p.H = heads; p.Hkv = kv_heads; p.q_len = q_len; p.N = kv_len; p.Npad = npad;          \
    p.causal = causal & 1; p.fast = (causal & SPATTEN_PREFILL_FAST_NUMERICS) != 0 && !scores && !col_importance && !lse; \
    p.sqrt_d = sqrtf((float)head_dim); p.nqb = ceil_div(q_len, 256);                                   \
    p.ksplit = (!scores && !col_importance && prefill_variant() == 0) ? flash_ksplit(batch, heads, q_len, kv_len) : 1; \
    p.part_o = (float*)(ws + align256((size_t)batch * kv_heads * head_dim * npad * 2));                \
    p.part_ml = p.part_o + (size_t)batch * heads * p.nqb * p.ksplit * 256 * head_dim;                   \
    p.pair = flash_pair(batch, heads, q_len, causal & 1) && p.ksplit == 1 && !scores && !col_importance && !mask && !p.fast && prefill_variant() == 0; \
    return launch_flash<T, DD>(p, st);                                                                 \
  }
  if (dtype == SPATTEN_BF16) { if (head_dim == 128) SPATTEN_FLASH(bf16_t, 128) else SPATTEN_FLASH(bf16_t, 64) }
  else { if (head_dim == 128) SPATTEN_FLASH(f16_t, 128) else SPATTEN_FLASH(f16_t, 64) }
#undef SPATTEN_FLASH
}

extern "C" size_t spatten_importance_prefill_workspace_bytes(int batch, int heads, int head_dim, int q_len) {
  if (batch <= 0 || heads <= 0 || head_dim <= 0 || q_len <= 0) return 0;
  return 256 + align256((size_t)batch * heads * q_len * head_dim * 2);
}

// Cascade (cumulative) importance of a multi-token forward, stash-free (README.md:11): acc[h, j] += sum_{b, i} softmax
// probability of key j for query row i, from the row statistics `lse` that spatten_attn_prefill wrote for the SAME
// inputs.  Two launches: rotate the queries into the workspace (spatten_rope_single's kernel), then the colprob kernel.
int rope_rows(int dtype, const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sn, void* y, int64_t y_sb, int64_t y_sh,
              int64_t y_sn, const void* cos, const void* sin, int table_rows, const int64_t* position_ids, int64_t pos_sb,
              int pos0, int batch, int heads, int n, int head_dim, void* stream);

extern "C" int spatten_importance_accumulate_prefill(int dtype, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq,
                                                     const void* kr_cache, int64_t kv_sb, int64_t kv_sh, const void* cos,
                                                     const void* sin, int table_rows, const int64_t* position_ids,
                                                     int64_t pos_sb, const float* lse, float* acc, int64_t acc_sh,
                                                     void* workspace, int batch, int heads, int kv_heads, int head_dim,
                                                     int q_len, int kv_len, int pos_q0, int causal, void* stream) {
  if (!q || !kr_cache || !cos || !sin || !lse || !acc || !workspace) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || q_len <= 0 || kv_len < q_len || pos_q0 < 0)
    return SPATTEN_ERR_INVALID;
  if ((dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) || (head_dim != 64 && head_dim != 128)) return SPATTEN_ERR_UNSUPPORTED;
  char* ws = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
  const int64_t qr_sh = (int64_t)q_len * head_dim, qr_sb = (int64_t)heads * qr_sh;
  int rc = rope_rows(dtype, q, q_sb, q_sh, q_sq, ws, qr_sb, qr_sh, head_dim, cos, sin, table_rows, position_ids, pos_sb, pos_q0,
                     batch, heads, q_len, head_dim, stream);
  if (rc != SPATTEN_OK) return rc;
  const dim3 grid((unsigned)ceil_div(kv_len, 256), (unsigned)heads, (unsigned)batch);
#define SPATTEN_COLPROB(T, DD)                                                                              \
  {                                                                                                         \
    ColProbParams<T> p;                                                                                     \
    p.qrot = (const T*)ws; p.q_sb = qr_sb; p.q_sh = qr_sh; p.kr = (const T*)kr_cache; p.kv_sb = kv_sb; p.kv_sh = kv_sh; \
    p.lse = lse; p.acc = acc; p.acc_sh = acc_sh; p.H = heads; p.Hkv = kv_heads; p.q_len = q_len; p.N = kv_len; \
    p.causal = causal & 1; p.sqrt_d = sqrtf((float)head_dim);                                               \
    hipLaunchKernelGGL((prefill_colprob_kernel<T, DD>), grid, dim3(512), 0, (hipStream_t)stream, p);         \
  }
  if (dtype == SPATTEN_BF16) { if (head_dim == 128) SPATTEN_COLPROB(bf16_t, 128) else SPATTEN_COLPROB(bf16_t, 64) }
  else { if (head_dim == 128) SPATTEN_COLPROB(f16_t, 128) else SPATTEN_COLPROB(f16_t, 64) }
#undef SPATTEN_COLPROB
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" size_t spatten_prefill_pq_workspace_bytes(int dtype, int batch, int heads, int kv_heads, int head_dim,
                                                     int q_len, int kv_len) {
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || head_dim <= 0 || q_len <= 0 || kv_len <= 0) return 0;
  const size_t npad = (size_t)ceil_div(kv_len, 128) * 128, rows = (size_t)batch * kv_heads * kv_len;
  return 256 + align256((size_t)batch * kv_heads * head_dim * npad * 2) + 2 * align256(rows * head_dim * 2) + align256(rows * 4)
         + align256((size_t)batch * heads * q_len * 4) + align256((size_t)batch * heads * 4)               // pass 2: row lists + counts
         + align256(flash_partial_bytes(batch, heads, head_dim, q_len, pq_pass2_ksplit(kv_len)));          // ... and its key-split partials
}

// Prefill over progressively quantised keys (BASELINE.json configs[3]): MSB-first fetch, max-probability decision PER
// QUERY ROW, LSB refetch + ONE recompute of the flagged rows (RequantDecision.scala:44-72; SpAttenController.scala:402
// applies the rule to every query row, not only to single-token steps).  Three launches + the V transpose:
//   expand   planes -> integer-valued keys in the model dtype (exact) + scale / sqrt(d) per key        (pq.hip)
//   pass 1   the 128-key-tile flash kernel over the MSB keys: logits = (q . 16 msb) * scale / sqrt(d) in fp32, softmax, P.V,
//            need_lsb[b,h,i] = max_j prob_ij < threshold
//   pass 2   the same kernel over the 8-bit keys, only for blocks / waves that hold a flagged row; flagged rows overwrite
//            their output
extern "C" int spatten_attn_prefill_pq(int dtype, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq, const void* msb,
                                       const void* lsb, const float* scale, int64_t pl_sb, int64_t pl_sh, int64_t sc_sb,
                                       int64_t sc_sh, const void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos,
                                       const void* sin, int table_rows, const int64_t* position_ids, int64_t pos_sb,
                                       const void* mask, int64_t mask_sb, int64_t mask_sq, void* out, int64_t out_sb,
                                       int64_t out_sq, int32_t* need_lsb, float threshold, void* workspace, int batch,
                                       int heads, int kv_heads, int head_dim, int q_len, int kv_len, int pos_q0,
                                       int causal, void* stream) {
  if (!q || !msb || !lsb || !scale || !v_cache || !cos || !sin || !out || !need_lsb || !workspace) return SPATTEN_ERR_INVALID;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || q_len <= 0 || kv_len < q_len || pos_q0 < 0)
    return SPATTEN_ERR_INVALID;
  if (threshold < 0.f) return SPATTEN_ERR_INVALID;
  if (table_rows < kv_len || (!position_ids && pos_q0 + q_len > table_rows)) return SPATTEN_ERR_INVALID;
  if ((dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) || (head_dim != 64 && head_dim != 128)) return SPATTEN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
  const int npad = ceil_div(kv_len, 128) * 128;
  const size_t rows = (size_t)batch * kv_heads * kv_len;
  void* vt = ws;
  char* k_msb = ws + align256((size_t)batch * kv_heads * head_dim * npad * 2);
  char* k_full = k_msb + align256(rows * head_dim * 2);
  float* kscale = (float*)(k_full + align256(rows * head_dim * 2));
  int32_t* row_list = (int32_t*)((char*)kscale + align256(rows * 4));
  int32_t* row_cnt = (int32_t*)((char*)row_list + align256((size_t)batch * heads * q_len * 4));
  const int ks2 = pq_pass2_ksplit(kv_len);
  float* part2 = (float*)((char*)row_cnt + align256((size_t)batch * heads * 4));
  {
    const dim3 grid((unsigned)(npad / 64), (unsigned)kv_heads, (unsigned)batch);
    if (head_dim == 128)
      hipLaunchKernelGGL((vt_kernel<128>), grid, dim3(256), 0, st, (const uint16_t*)v_cache, kv_sb, kv_sh, (uint16_t*)vt, kv_len, npad, kv_heads);
    else
      hipLaunchKernelGGL((vt_kernel<64>), grid, dim3(256), 0, st, (const uint16_t*)v_cache, kv_sb, kv_sh, (uint16_t*)vt, kv_len, npad, kv_heads);
    if (hipGetLastError() != hipSuccess) return SPATTEN_ERR_LAUNCH;
  }
  int rc = pq_expand(dtype, msb, lsb, scale, pl_sb, pl_sh, sc_sb, sc_sh, k_msb, k_full, kscale, batch, kv_heads, head_dim, kv_len, st);
  if (rc != SPATTEN_OK) return rc;
#define SPATTEN_FLASH_PQ(T, DD, KPTR, THR)                                                              \
  {                                                                                                    \
    FlashParams<T> p;                                                                                  \
    p.rows = (THR) < 0.f ? row_list : nullptr; p.row_cnt = row_cnt;                                    \
    p.q = (const T*)q; p.q_sb = q_sb; p.q_sh = q_sh; p.q_sq = q_sq;                                    \
    p.cos = (const T*)cos; p.sin = (const T*)sin; p.table_rows = table_rows;                           \
    p.pos_ids = position_ids; p.pos_sb = pos_sb; p.pos_q0 = pos_q0;                                    \
    p.kr = (const T*)(KPTR); p.kv_sb = (int64_t)kv_heads * kv_len * head_dim; p.kv_sh = (int64_t)kv_len * head_dim; \
    p.vt = (const T*)vt; p.mask = (const T*)mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;            \
    p.v = (const T*)v_cache; p.v_sb = kv_sb; p.v_sh = kv_sh; p.vtr = 0;                                \
    p.out = (T*)out; p.out_sb = out_sb; p.out_sq = out_sq;                                             \
    p.scores = nullptr; p.sc_sb = p.sc_sh = p.sc_sq = 0; p.col_imp = nullptr; p.lse = nullptr;         \
    p.kscale = kscale; p.ks_sb = (int64_t)kv_heads * kv_len; p.ks_sh = kv_len; p.need = need_lsb; p.pq_thr = (THR); \
    p.B = batch; p.H = heads; p.Hkv = kv_heads; p.q_len = q_len; p.N = kv_len; p.Npad = npad;          \
    p.causal = causal & 1; p.fast = 0; p.sqrt_d = sqrtf((float)head_dim); p.nqb = ceil_div(q_len, 256); p.pair = 0;   \
    p.ksplit = 1; p.part_o = nullptr; p.part_ml = nullptr;                                             \
    if ((THR) < 0.f && ks2 > 1) {                                                                      \
      p.ksplit = ks2; p.part_o = part2;                                                                \
      p.part_ml = part2 + (size_t)batch * heads * p.nqb * ks2 * 256 * head_dim;                        \
    }                                                                                                  \
    rc = launch_flash<T, DD>(p, st);                                                                   \
  }
#define SPATTEN_FLASH_PQ_ANY(KPTR, THR)                                                                                  \
  if (dtype == SPATTEN_BF16) { if (head_dim == 128) SPATTEN_FLASH_PQ(bf16_t, 128, KPTR, THR) else SPATTEN_FLASH_PQ(bf16_t, 64, KPTR, THR) } \
  else { if (head_dim == 128) SPATTEN_FLASH_PQ(f16_t, 128, KPTR, THR) else SPATTEN_FLASH_PQ(f16_t, 64, KPTR, THR) }
  SPATTEN_FLASH_PQ_ANY(k_msb, threshold)          // pass 1 (pq_thr >= 0 selects it)
  if (rc != SPATTEN_OK) return rc;
  hipLaunchKernelGGL(pq_rows_compact_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0, st, need_lsb, row_list, row_cnt, q_len);
  if (hipGetLastError() != hipSuccess) return SPATTEN_ERR_LAUNCH;
  SPATTEN_FLASH_PQ_ANY(k_full, -1.0f)              // pass 2: flagged rows from the 8-bit keys
#undef SPATTEN_FLASH_PQ_ANY
#undef SPATTEN_FLASH_PQ
  return rc;
}

// developer diagnostic (not part of the boundary, not in include/spatten.h): which flash kernel the last prefill call launched

#ifdef SPATTEN_PF_TRACE
extern "C" int spatten_debug_set_pf_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pf_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
