// decode_gqa.hip — the single-row decode step of a GROUPED-QUERY model (modify_llama.py:86-147 with repeat_kv, :106-108: H query
// heads on Hkv cached heads, G = H / Hkv) on the matrix cores (round 6).
//
// The general decode kernel (decode_body.h) launches one workgroup column per QUERY head: the G heads of a group each stream their
// kv head's rows — G x the unique bytes (r05: 2.2 TB/s of unique K / V at 16384 rows, 32 / 8 heads).  Here a workgroup column is a
// KV head: its rows are streamed ONCE and scored for the whole group with the flash kernel's products (prefill_attn.hip):
//     S^T = Kr · Qrot^T   (32 keys x 32 columns per v_mfma_f32_32x32x16, the group's query heads are the first G columns)
//     O^T = Vt · P^T      (d x columns, the contraction index = keys)
// so a lane owns one query head (column) and 16 of a tile's 32 keys; the online softmax needs one lane <-> lane + 32 exchange per
// tile.  The matrix pipe runs at a few per cent — the kernel is a STREAM: what matters is bytes in flight.
//
// Mapping: grid = (S splits, Hkv, B), 256 threads = 4 waves, ONE wave per SIMD, one workgroup per CU (128 KiB of LDS).  A split owns
// `chunk` consecutive rows; its 32-row tiles go round-robin to the four waves, which never synchronise inside the stream: every wave
// has a private two-stage LDS ring of { K tile, V tile } (8 + 8 KiB) filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
// registers; non-temporal): a stage is refilled with the tile after next in two halves — its keys behind S, its values behind P·V — so one to
// two tiles (16-32 KiB) per wave are in flight.  A split's chunk is at least one tile per wave (128 rows).  K tiles
// are XOR-swizzled on the global side for conflict-free ds_read_b128 A-operands; V tiles stay row-major and are read with gfx950's
// transposing ds_read_b64_tr_b16 (the contraction index of P·V is the key = the row).
// The appended token (k_new / v_new, modify_llama.py:95-100) is written INTO the LDS tile that holds row N-1 (and to the caches), so
// the tile loop has no special row.  Logits carry both reference roundings (:111-113); the stash holds them pre-mask (:116-119).
// Split partials ({value, tag} granules) and the merge are the decode family's (decode_body.h): the workspace, its generation words
// and the error word are shared with the per-query-head kernels, so launches of both kinds may alternate on one workspace.
#include <stdlib.h>

#include "common.h"
#include "decode_body.h"
#include "mfma_tiles.h"

#ifdef SPATTEN_GQA_TRACE   // developer instrumentation (tools/mb/gqa_trace.py): per (workgroup, wave) phase stamps, device-wide 100 MHz clock
static __device__ unsigned long long* g_gqa_trace = nullptr;
#define SPATTEN_GSTAMP(slot)                                                                                          \
  do {                                                                                                                \
    if (g_gqa_trace && lane == 0)                                                                                     \
      g_gqa_trace[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (tid >> 6)) * 48 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define SPATTEN_GSTAMP(slot)
#endif
// per tile k < 4 of a wave: 0 keys ready, 1 S done, 2 key refill issued, 3 softmax done, 4 values ready, 5 P.V done, 6 value refill issued
#define SPATTEN_GSTAMP_IT(k, j) do { if ((k) < 4) SPATTEN_GSTAMP(12 + 8 * (k) + (j)); } while (0)

#ifndef SPATTEN_GQA_NT             // A/B: non-temporal LDS-DMA requests
#define SPATTEN_GQA_NT 1
#endif
#ifndef SPATTEN_GQA_SPLIT_REFILL   // A/B: a consumed stage is refilled in two halves (keys behind S, values behind P·V)
#define SPATTEN_GQA_SPLIT_REFILL 1
#endif

namespace spatten {

template <typename T>
struct GqaParams {
  const T* q; int64_t q_sb, q_sh;
  T* kc; T* krc; T* vc; int64_t kv_sb, kv_sh;
  const T* k_new; const T* v_new; int64_t new_sb, new_sh; int append;
  const T* cos; const T* sin; int table_rows, nr_row;      // rotary half tables [rows, D/2]; nr_row: the appended slot's row
  const int64_t* pos_ids; int64_t pos_sb; int pos_q;
  const int32_t* step;                                      // device-length form: word 0 = the live length (N is then the bound)
  T* out; int64_t out_sb;
  T* scores; int64_t sc_sb, sc_sh; int sc_vec;              // sc_vec: stash rows 8-byte aligned -> four logits per store; 2: 16-byte aligned -> eight
  float* lse;
  unsigned* ws_err; unsigned* ws_cnt; unsigned long long* ws_part; int64_t ws_unit;
  int B, H, Hkv, G, N, S, chunk, poll_merge;
  float sqrt_d;
};

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt / lgkmcnt left at "no wait")
template <int N> __device__ inline void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

template <typename T, bool DYN>
__global__ __launch_bounds__(256, 1) void decode_gqa_kernel(const GqaParams<T> p) {
  constexpr int D = 128, KK = D / 16, DB = D / 32, KT = 32, NW = 4;
  constexpr int KBYTES = KT * D * 2;                  // a tile's key rows (8 KiB); as many value rows
  constexpr int STAGE = 2 * KBYTES, WBUF = 2 * STAGE; // a wave's ring: two stages of { K, V }
  constexpr int NK = KBYTES / 1024;                   // DMA instructions per tile and operand
  constexpr int PITCH = D + 4;                        // floats per (wave, column) partial in LDS: o[D], m, l
  using frag = typename Mfma<T>::frag;
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  __shared__ __attribute__((aligned(1024))) char lds[NW * WBUF];
  __shared__ unsigned s_gen[32];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const int G = p.G;
  const int unit0 = b * p.H + hkv * G;                // the group's first softmax row (workspace index of head hkv * G)
  const int lo = split * p.chunk;
  char* const wbuf = lds + wave * WBUF;
  SPATTEN_GSTAMP(0);

  int n_live = p.N;
  if (DYN) n_live = p.step[0];                        // (a scalar load: waited for behind the first tile's requests)

  const T* krb = p.krc + b * p.kv_sb + hkv * p.kv_sh;
  const T* vb = p.vc + b * p.kv_sb + hkv * p.kv_sh;
  const int64_t plane_bytes = (int64_t)p.N * D * 2;   // rows past the bound read as zeros
  // This wave's 1-KiB pieces of a tile (4 rows each).  Keys: lane (row-in-piece lr, physical slot ps) fetches logical slot
  // ps ^ (row & 15); values: ps ^ 4 (row & 3) (prefill_attn.hip: the transposing reads' bank layout).
  // The requests are INLINE ASSEMBLY: behind the compiler's own LDS-DMA builtin every later LDS read gets an `s_waitcnt vmcnt(0)` in
  // front of it (the two may alias for all it knows) — which would also wait for the tile requested a moment ago, i.e. halve what a
  // wave keeps in flight.  The kernel's own vmcnt waits (below) order the ring.  (m0 is used by nothing else in this kernel.)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto rsrc_of = [&](const T* base) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)__builtin_amdgcn_readfirstlane((uint32_t)a), (int)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xFFFFu),
                 (int)(plane_bytes < 0x7FFFFFFF ? plane_bytes : 0x7FFFFFFF), 0x00020000};
  };
  const i32x4 k_rsrc = rsrc_of(krb), v_rsrc = rsrc_of(vb);
  const unsigned lds_wbuf = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (unsigned)wave * WBUF;
  auto dma16a = [&](const i32x4& rsrc, unsigned lds_dst, int voff, int soff) {
#if SPATTEN_GQA_NT      // non-temporal: every K / V byte is used once per launch
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
  };
  auto dma_keys = [&](int tile, int stage) {
    const int ln = opaque_lane(lane);
    const int lr = ln >> 4, ps = ln & 15;
    const int kbase = lr * 256 + ((ps ^ lr) << 4);
    const unsigned ka = lds_wbuf + stage * STAGE;
    const int soff = __builtin_amdgcn_readfirstlane((lo + tile * KT) * (D * 2));
#pragma unroll
    for (int i = 0; i < NK; ++i) dma16a(k_rsrc, ka + i * 1024, kbase ^ (((i * 4) & 15) << 4), soff + i * 1024);
  };
  auto dma_values = [&](int tile, int stage) {
    const int ln = opaque_lane(lane);
    const int lr = ln >> 4, ps = ln & 15;
    const int vbase = lr * 256 + ((ps ^ (4 * lr)) << 4);
    const unsigned va = lds_wbuf + stage * STAGE + KBYTES;
    const int soff = __builtin_amdgcn_readfirstlane((lo + tile * KT) * (D * 2));
#pragma unroll
    for (int i = 0; i < NK; ++i) dma16a(v_rsrc, va + i * 1024, vbase, soff + i * 1024);
  };
  auto dma_tile = [&](int tile, int stage) { dma_keys(tile, stage); dma_values(tile, stage); };
  // ---- the group's queries as B operands: column g = head hkv G + g, fragment kk holds elements [16 kk + 8 hi, +8), so kk and
  // kk + KK/2 are the (x[i], x[i + d/2]) pairs RoPE combines.  Columns >= G: zeros.  Their rows are requested FIRST (a wave's loads
  // return in order: behind the tile the rotation would wait for the tile), then the first tile, then the rotation.
  raw_t q_raw[KK / 2][4];
  {
    const int gq = min(col, G - 1);
    const T* qrow = p.q + b * p.q_sb + (int64_t)(hkv * G + gq) * p.q_sh;
    int ps = DYN ? 0 : (p.pos_ids ? (int)p.pos_ids[b * p.pos_sb] : p.pos_q);
    ps = min(max(ps, 0), p.table_rows - 1);
    const T* cr = p.cos + (int64_t)ps * (D / 2);
    const T* sr = p.sin + (int64_t)ps * (D / 2);
#pragma unroll
    for (int kk = 0; kk < KK / 2; ++kk) {
      q_raw[kk][0] = V8::ldg(qrow + 16 * kk + 8 * hi);
      q_raw[kk][1] = V8::ldg(qrow + D / 2 + 16 * kk + 8 * hi);
      q_raw[kk][2] = V8::ldg(cr + 16 * kk + 8 * hi);
      q_raw[kk][3] = V8::ldg(sr + 16 * kk + 8 * hi);
    }
  }
  // launch generations of the group's units (tags of the partials), parked in LDS for the publish / merge loops
  if (p.S > 1 && tid < G) s_gen[tid] = p.ws_cnt[2 * (unit0 + tid) + 1];
  __builtin_amdgcn_sched_barrier(0);
  // the first tile goes out before anything is waited for (laid out for the bound: a device-length step may find it past its length)
  const int nt_bound = (min(lo + p.chunk, p.N) - lo + KT - 1) / KT;
  if (wave < nt_bound) dma_tile(wave, 0);
  if (wave + NW < nt_bound) dma_tile(wave + NW, 1);
  SPATTEN_GSTAMP(1);          // two tiles requested
  __builtin_amdgcn_sched_barrier(0);
  frag qf[KK];                                        // rotated (modify_llama.py:92: three rounded ops)
  {
    const float live = col < G ? 1.f : 0.f;
#pragma unroll
    for (int kk = 0; kk < KK / 2; ++kk) {
      float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
      V8::unpack(q_raw[kk][0], xlo);
      V8::unpack(q_raw[kk][1], xhi);
      V8::unpack(q_raw[kk][2], cc);
      V8::unpack(q_raw[kk][3], ss);
      rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[kk][e] = DT<T>::from_f32(ylo[e] * live); qf[kk + KK / 2][e] = DT<T>::from_f32(yhi[e] * live); }
    }
  }

  // ---- the live rows of this split -----------------------------------------------------------------------------------------------
  const int N = DYN ? min(__builtin_amdgcn_readfirstlane(n_live), p.N) : p.N;
  const int hi_all = min(lo + p.chunk, N);
  const int nt = hi_all > lo ? (hi_all - lo + KT - 1) / KT : 0;      // tiles of this split that hold a live row
  const int n_cached = p.append ? N - 1 : N;                         // rows the caches hold already
  SPATTEN_GSTAMP(2);          // queries rotated, length known
  // the wave whose last tile holds row N-1 appends the new token: its rows are requested now and used in that tile
  const bool owns_new = p.append && lo < N && hi_all == N && ((nt - 1) & (NW - 1)) == wave;
  raw_t nk_raw[4], nv_raw;
  if (owns_new) {
    const int c8 = lane & 7;
    const T* kp = p.k_new + b * p.new_sb + hkv * p.new_sh;
    const int nr = DYN ? 1 : p.nr_row;
    nk_raw[0] = V8::ldg(kp + 8 * c8);
    nk_raw[1] = V8::ldg(kp + D / 2 + 8 * c8);
    nk_raw[2] = V8::ldg(p.cos + (int64_t)nr * (D / 2) + 8 * c8);
    nk_raw[3] = V8::ldg(p.sin + (int64_t)nr * (D / 2) + 8 * c8);
    nv_raw = V8::ldg(p.v_new + b * p.new_sb + hkv * p.new_sh + 8 * (lane & 15));
  }

  const float rsqrt_d = 1.0f / p.sqrt_d;
  T* stashp = (p.scores != nullptr && col < G) ? p.scores + b * p.sc_sb + (int64_t)(hkv * G + col) * p.sc_sh : nullptr;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  const unsigned lds0 = (unsigned)(wbuf - lds);
  const int li = lane & 15, kr = li >> 2;
  for (int i = wave, k = 0; i < nt; i += NW, ++k) {
    const int stage = k & 1;
    const bool more = i + NW < nt;                    // this wave's next tile is in flight behind this one
    const bool more2 = i + 2 * NW < nt;               // ... and the tile after next is requested during this one
    const int row0 = lo + i * KT;
    const bool edge = row0 + KT > n_cached;           // the tile holds rows the caches do not: past the length, or the new token's
    // loads return in order: "at most X outstanding" with X = the requests issued behind the wanted ones proves those have landed
    // whatever the stash stores in between do (they may only make the wait longer)
    if (edge) {
      if (more) wait_vm<2 * NK>(); else wait_vm<0>();
      // value rows the cache does not hold: zero (their P is 0, but 0 x stale bits may be NaN); then the new token's rows
      char* va = wbuf + stage * STAGE + KBYTES;
#pragma unroll
      for (int j = 0; j < KBYTES / 16 / 64; ++j) {
        const int slot = lane + 64 * j;
        if (row0 + (slot >> 4) >= n_cached) *reinterpret_cast<u32x4*>(va + slot * 16) = u32x4{0u, 0u, 0u, 0u};
      }
      if (owns_new && row0 + KT > N - 1 && row0 <= N - 1) {
        const int jn = N - 1, rr = jn - row0;
        float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
        V8::unpack(nk_raw[0], xlo);
        V8::unpack(nk_raw[1], xhi);
        V8::unpack(nk_raw[2], cc);
        V8::unpack(nk_raw[3], ss);
        rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
        const raw_t r_lo = V8::pack(ylo), r_hi = V8::pack(yhi);
        char* ka = wbuf + stage * STAGE;
        const int c8 = lane & 7;
        if (lane < 8) {      // rotated key: logical slots c8 (elements 8 c8 ..) and 8 + c8 of row rr
          *reinterpret_cast<raw_t*>(ka + rr * 256 + ((c8 ^ (rr & 15)) << 4)) = r_lo;
          *reinterpret_cast<raw_t*>(ka + rr * 256 + (((8 + c8) ^ (rr & 15)) << 4)) = r_hi;
          T* krow = p.krc + b * p.kv_sb + hkv * p.kv_sh + (int64_t)jn * D;
          V8::stg(krow + 8 * c8, r_lo);
          V8::stg(krow + D / 2 + 8 * c8, r_hi);
          if (p.kc != nullptr) {                      // un-rotated into the key cache (modify_llama.py:95-100)
            T* krow0 = p.kc + b * p.kv_sb + hkv * p.kv_sh + (int64_t)jn * D;
            V8::stg(krow0 + 8 * c8, nk_raw[0]);
            V8::stg(krow0 + D / 2 + 8 * c8, nk_raw[1]);
          }
        }
        if (lane < 16) {
          *reinterpret_cast<raw_t*>(va + rr * 256 + (((lane & 15) ^ (4 * (rr & 3))) << 4)) = nv_raw;
          V8::stg(p.vc + b * p.kv_sb + hkv * p.kv_sh + (int64_t)jn * D + 8 * (lane & 15), nv_raw);
        }
      }
    } else {
      if (more) wait_vm<3 * NK>(); else wait_vm<NK>();            // the tile's key rows have landed
    }
    __builtin_amdgcn_sched_barrier(0);
    if (k == 0) SPATTEN_GSTAMP(3);   // first tile's keys landed
    if (k == 1) SPATTEN_GSTAMP(5);   // second tile's keys landed
    SPATTEN_GSTAMP_IT(k, 0);

    // ---- S^T = Kr · Qrot^T ------------------------------------------------------------------------------------------------------
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const unsigned ku = lds0 + stage * STAGE + col * 256 + ((col & 15) << 4);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const frag a = *reinterpret_cast<const frag*>(lds + (ku ^ ((2 * kk + hi) << 4)));
        s = Mfma<T>::mma(a, qf[kk], s);
      }
    }
    SPATTEN_GSTAMP_IT(k, 1);
#if SPATTEN_GQA_SPLIT_REFILL
    // the stage's KEY half is consumed (the reads that fed the products above have returned): the keys of the tile after next go
    // out now, a softmax and a P·V earlier than the stage's value half — a wave's requests then enter the CU's memory queue in two
    // bursts of 8 KiB per tile instead of one of 16
    __builtin_amdgcn_sched_barrier(0);
    if (more2) dma_keys(i + 2 * NW, stage);
    __builtin_amdgcn_sched_barrier(0);
#endif
    SPATTEN_GSTAMP_IT(k, 2);
    // both reference roundings of every logit (matmul -> dtype, "/ sqrt(d)" -> dtype, modify_llama.py:111-113)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 x = round2<T>(f32x2{s[r], s[r + 1]});
      const f32x2 v = round2<T>(f32x2{logit_scale<T>(x[0], p.sqrt_d, rsqrt_d), logit_scale<T>(x[1], p.sqrt_d, rsqrt_d)});
      s[r] = v[0];
      s[r + 1] = v[1];
    }
    // the stash (pre-mask logits, :116-119): register r is key row0 + (r & 3) + 8 (r >> 2) + 4 hi
    if (stashp != nullptr && p.sc_vec == 2 && row0 + KT <= N) {
      // a whole tile, 16-byte aligned rows: lanes col and col + 32 exchange halves (v_permlane32_swap: 4 per tile) so that each
      // holds 16 CONSECUTIVE keys — two 16-byte stores per tile instead of four 8-byte ones (a store takes a slot of the CU's
      // memory queue like a 1-KiB request does)
      uint32_t w[4][2];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const T x0 = DT<T>::from_f32(s[4 * q4 + 2 * j]), x1 = DT<T>::from_f32(s[4 * q4 + 2 * j + 1]);
          w[q4][j] = (uint32_t)*reinterpret_cast<const unsigned short*>(&x0) | ((uint32_t)*reinterpret_cast<const unsigned short*>(&x1) << 16);
        }
      u32x4 lo8, hi8;                                 // keys [16 hi, +8) and [16 hi + 8, +8) of the tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const auto ra = __builtin_amdgcn_permlane32_swap(w[0][j], w[2][j], false, false);
        lo8[j] = ra[0]; lo8[2 + j] = ra[1];
        const auto rb = __builtin_amdgcn_permlane32_swap(w[1][j], w[3][j], false, false);
        hi8[j] = rb[0]; hi8[2 + j] = rb[1];
      }
      T* dst = stashp + row0 + 16 * hi;
      *reinterpret_cast<u32x4*>(dst) = lo8;
      *reinterpret_cast<u32x4*>(dst + 8) = hi8;
    } else if (stashp != nullptr) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int key = row0 + 8 * q4 + 4 * hi;
        if (p.sc_vec && key + 3 < N) {
          typedef T t4 __attribute__((ext_vector_type(4)));
          t4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = DT<T>::from_f32(s[4 * q4 + e]);
          *reinterpret_cast<t4*>(stashp + key) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (key + e < N) stashp[key + e] = DT<T>::from_f32(s[4 * q4 + e]);
        }
      }
    }
    if (row0 + KT > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        s[r] = key < N ? s[r] : -INFINITY;
      }
    }
    // ---- online softmax of the column (lanes col and col + 32 hold its 32 keys) ---------------------------------------------------
    float mt = fmaxf(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mt = fmaxf(mt, fmaxf(s[r], s[r + 1]));
    mt = xor32_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float mu = (m_new == -INFINITY) ? 0.f : m_new;
    if (m_new > m_run) {                              // the first tile; rare afterwards
      const float alpha = __expf(m_run - mu);         // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      m_run = m_new;
    }
    frag pf[2];
    {
      constexpr float kLog2e = 1.4426950408889634f;
      const float m2 = mu * kLog2e;
      float ls[2] = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[t * 8 + e], kLog2e, -m2));     // exp2(-inf) = 0
          ls[e & 1] += pv;
          pf[t][e] = DT<T>::from_f32(pv);
        }
      l_run += ls[0] + ls[1];
    }
    SPATTEN_GSTAMP_IT(k, 3);
    // ---- O^T += Vt · P^T: the tile's value rows have landed ------------------------------------------------------------------------
    __builtin_amdgcn_sched_barrier(0);
#if SPATTEN_GQA_SPLIT_REFILL
    if (!edge) { if (more2) wait_vm<3 * NK>(); else if (more) wait_vm<2 * NK>(); else wait_vm<0>(); }   // (behind V: K, V of the next tile, K of the one after)
#else
    if (!edge) { if (more) wait_vm<2 * NK>(); else wait_vm<0>(); }
#endif
    __builtin_amdgcn_sched_barrier(0);
    SPATTEN_GSTAMP_IT(k, 4);
    {
      // lane (col, hi) needs, for d = 32 db + col, the 8 keys its P fragment holds — elements 0..3: keys 16 t + 4 hi + 0..3,
      // elements 4..7: 16 t + 8 + 4 hi + 0..3 — two transposing reads 8 rows apart (prefill_attn.hip, the VTR form)
      typedef short v4s __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) v4s* lds_v4s;
      const unsigned vtu = lds0 + stage * STAGE + KBYTES + (4 * hi + kr) * 256 + ((lane >> 4) & 1) * 32 + ((li & 3) >> 1) * 16 + (li & 1) * 8;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const unsigned a0 = vtu + (t * 16) * 256 + (((unsigned)db ^ (unsigned)kr) << 6);
          union { v4s h[2]; frag f; } u;
          u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + a0));
          u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + a0 + 2048));
          o[db] = Mfma<T>::mma(u.f, pf[t], o[db]);
        }
    }
    // the stage is consumed (its last read fed the last product): request the tile after next into it
    __builtin_amdgcn_sched_barrier(0);
    if (k == 0) SPATTEN_GSTAMP(4);   // first tile consumed
    SPATTEN_GSTAMP_IT(k, 5);
#if SPATTEN_GQA_SPLIT_REFILL
    if (more2) dma_values(i + 2 * NW, stage);
#else
    if (more2) dma_tile(i + 2 * NW, stage);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (k == 0) SPATTEN_GSTAMP(6);   // its refill requested
    SPATTEN_GSTAMP_IT(k, 6);
  }

  // ---- the four waves' partials -> LDS (each wave into its own ring: all its requests have landed and been consumed) ----------------
  {
    wait_vm<0>();                                     // (a device-length step may have requested a tile it then found past its length)
    SPATTEN_GSTAMP(7);        // stream done
    const float l_col = xor32_sum(l_run);
    float* part = reinterpret_cast<float*>(wbuf);
    if (col < G) {
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<f32x4*>(part + col * PITCH + 32 * db + 8 * q4 + 4 * hi) =
              f32x4{o[db][4 * q4], o[db][4 * q4 + 1], o[db][4 * q4 + 2], o[db][4 * q4 + 3]};
      if (hi == 0) { part[col * PITCH + D] = m_run; part[col * PITCH + D + 1] = l_col; }
    }
  }
  __syncthreads();
  SPATTEN_GSTAMP(8);          // all waves' partials in LDS

  // ---- fold the waves, publish the split's partial per head ---------------------------------------------------------------------------
  const int items = G * D;                            // (head of the group, output element)
  for (int idx = tid; idx < items; idx += 256) {
    const int g = idx >> 7, e = idx & (D - 1);
    float mw[NW], mwg = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      mw[w] = reinterpret_cast<const float*>(lds + w * WBUF)[g * PITCH + D];
      mwg = fmaxf(mwg, mw[w]);
    }
    const float mu = (mwg == -INFINITY) ? 0.f : mwg;
    float o_tot = 0.f, l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* pw = reinterpret_cast<const float*>(lds + w * WBUF) + g * PITCH;
      const float ww = __expf(mw[w] - mu);            // exp(-inf) = 0: a wave without tiles
      o_tot = fmaf(pw[e], ww, o_tot);
      l_tot = fmaf(pw[D + 1], ww, l_tot);
    }
    const int unit = unit0 + g;
    if (p.S == 1) {
      p.out[b * p.out_sb + (int64_t)(hkv * G + g) * D + e] = DT<T>::from_f32(o_tot / l_tot);
      if (p.lse != nullptr && e == 0) { p.lse[2 * unit] = mwg; p.lse[2 * unit + 1] = l_tot; }
      continue;
    }
    unsigned long long* part = p.ws_part + (int64_t)unit * p.ws_unit + (int64_t)split * (D + 2);
    const unsigned tag = (s_gen[g] & 0x7FFFFFFFu) + 1u;
    store_granule(part + e, o_tot, tag);
    if (e == 0) { store_granule(part + D, mwg, tag); store_granule(part + D + 1, l_tot, tag); }
  }
  if (p.S == 1) return;
  SPATTEN_GSTAMP(9);          // partial published (issued)

  // ---- merge: head g of the group is folded by ONE workgroup — all 256 threads: thread (e, half) takes every second split, 16 per
  // round trip (the granules and the (m, l) of a split: 48 independent loads), the halves meet through LDS.  Who merges: with the
  // whole grid co-resident (poll_merge) the workgroup of split S-1 - g % S simply polls for the other splits' granules — the G
  // heads of a group are merged by G different workgroups at once; otherwise (a polling workgroup could wait for one that cannot
  // start) the group's last ARRIVER merges all of them: its ticket tells it that every split has issued its granules.
  if (!p.poll_merge) {
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.ws_cnt + 2 * unit0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(p.S - 1)) return;
  }
  float* const fold = reinterpret_cast<float*>(lds);  // [D][3] of the upper half (the wave partials are dead behind the barrier below)
  constexpr int KB = 16;
  for (int g = 0; g < G; ++g) {
    if (p.poll_merge && split != p.S - 1 - (g % p.S)) continue;          // (workgroup-uniform)
    const int e = tid & (D - 1), half = tid >> 7;
    const int unit = unit0 + g;
    const unsigned long long* ws = p.ws_part + (int64_t)unit * p.ws_unit;
    const unsigned gen = s_gen[g];
    const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;
    float mg = -INFINITY, lg = 0.f, og = 0.f;
    bool expired = false;
    for (int s0 = half; s0 < p.S; s0 += 2 * KB) {
      unsigned long long ga[KB], gm[KB], gl[KB];
      int spins = 0;
      bool landed;
      do {     // every load is issued before any tag is looked at: one round trip per batch
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const int sc_ = (s0 + 2 * j) < p.S ? (s0 + 2 * j) : s0;
          const unsigned long long* qq = ws + (int64_t)sc_ * (D + 2);
          ga[j] = __hip_atomic_load(qq + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gm[j] = __hip_atomic_load(qq + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gl[j] = __hip_atomic_load(qq + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned diff = 0u;
#pragma unroll
        for (int j = 0; j < KB; ++j)
          diff |= ((unsigned)(ga[j] >> 32) ^ tag) | ((unsigned)(gm[j] >> 32) ^ tag) | ((unsigned)(gl[j] >> 32) ^ tag);
        landed = diff == 0u;
      } while (!landed && ++spins < (1 << 16));       // bounded: a granule that was issued always lands — if not, fail loudly
      expired |= !landed;
      float a[KB], ms[KB], ls[KB];
      float mn = mg;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const bool live = (s0 + 2 * j) < p.S;
        a[j] = live ? __uint_as_float((unsigned)ga[j]) : 0.f;
        ms[j] = live ? __uint_as_float((unsigned)gm[j]) : -INFINITY;
        ls[j] = live ? __uint_as_float((unsigned)gl[j]) : 0.f;
        mn = fmaxf(mn, ms[j]);
      }
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu);
      og *= w0; lg *= w0;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const float w = __expf(ms[j] - mu);
        og = fmaf(a[j], w, og);
        lg = fmaf(ls[j], w, lg);
      }
      mg = mn;
    }
    if (expired) {   // never merge incomplete data silently: flag the workspace and poison this unit's output
      atomicOr(p.ws_err, 1u);
      og = __builtin_nanf("");
    }
    SPATTEN_GSTAMP(10);       // the head's partials landed
    __syncthreads();                                  // the previous user of `fold` (wave partials / the last head) is done
    if (half == 1) { fold[3 * e] = og; fold[3 * e + 1] = mg; fold[3 * e + 2] = lg; }
    __syncthreads();
    if (half == 0) {
      const float m1 = fold[3 * e + 1];
      const float mn = fmaxf(mg, m1);
      const float mu = (mn == -INFINITY) ? 0.f : mn;
      const float w0 = __expf(mg - mu), w1 = __expf(m1 - mu);
      og = fmaf(fold[3 * e], w1, og * w0);
      lg = fmaf(fold[3 * e + 2], w1, lg * w0);
      p.out[b * p.out_sb + (int64_t)(hkv * G + g) * D + e] = DT<T>::from_f32(og / lg);
      if (e == 0) {
        if (p.lse != nullptr) { p.lse[2 * unit] = mn; p.lse[2 * unit + 1] = lg; }
        p.ws_cnt[2 * unit + 1] = gen + 1u;            // next launch: new tag
      }
    }
    SPATTEN_GSTAMP(11);       // merged head stored
  }
  if (!p.poll_merge && tid == 0) __hip_atomic_store(p.ws_cnt + 2 * unit0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
}

// 0 = never, 1 = whenever the launch is eligible, -1 (default) = eligible launches where the form measured faster (gqa_pays)
static std::atomic<int> g_gqa_mode{-2};
static int gqa_mode() {
  int m = g_gqa_mode.load(std::memory_order_relaxed);
  if (m == -2) {
    const char* e = getenv("SPATTEN_DECODE_GQA");
    m = e ? atoi(e) : -1;
    if (m < -1 || m > 1) m = -1;
    g_gqa_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
// Default mode: the matrix-core form where it measured faster than one workgroup column per query head (tools/mb/gqa_bench.py,
// 12 + 15 shapes, round 6).  The per-query-head step costs ~6.3 us + 0.047 us per 1024 (query head x row); this one a fixed part
// + 0.0625 us per 1024 (kv head x row): its slope does not grow with the group, its fixed part is the larger one — ~9.4 us while
// a split's chunk is one tile per wave (128 rows: short caches spread over many splits), ~11.6 us from two tiles per wave on
// (four waves' rings to fill twice over, the longer fold).  32 / 8 heads: from ~3k rows (4096: 11.6 against 12.4; 2048: 10.45
// against 9.44), 64 / 8 from ~1.5k (2048: 11.1 against 12.2), 16 / 8 only around 16k.
static inline bool gqa_pays(int batch, int heads, int kv_heads, int rows, int chunk) {
  const float n = (float)rows * (1.0f / 1024.0f) * (float)batch;
  const float per_query_head = 6.3f + 0.047f * (float)heads * n;
  const float fixed = chunk <= 128 ? 9.4f : chunk >= 256 ? 11.6f : 9.4f + 2.2f * (float)(chunk - 128) * (1.0f / 128.0f);
  return fixed + 0.0625f * (float)kv_heads * n < per_query_head;
}

// The split layout of a launch: one workgroup per CU (128 KiB of LDS each); a split gets at least one tile per wave (128 rows: 2048
// rows x 32 / 8 heads 12.6 -> 10.45 us, 4096 rows 13.2 -> 11.6 against two tiles per wave; 64 rows: as 128); chunks are whole tiles.
static inline void gqa_layout(int cols, int lay, int n_splits, int ws_splits, int& S, int& chunk32) {
  S = n_splits > 0 ? n_splits : std::max(1, coresident_workgroups() / cols);
  static int min_chunk = -1;                  // (SPATTEN_GQA_MIN_CHUNK: A/B of the shortest chunk a split may get)
  if (min_chunk < 0) { const char* e = getenv("SPATTEN_GQA_MIN_CHUNK"); min_chunk = e ? std::max(32, atoi(e)) : 128; }
  S = std::min(S, std::max(1, lay / min_chunk));
  S = std::min(S, std::min(ws_splits, kDecodeMaxSplits));
  const int chunk = (lay + S - 1) / S;
  chunk32 = (chunk + 31) / 32 * 32;
  S = (lay + chunk32 - 1) / chunk32;
}

// The grouped-query step on the matrix cores, when the launch is one this kernel serves: SPATTEN_OK after launching,
// SPATTEN_ERR_UNSUPPORTED (nothing launched: the caller takes the per-query-head kernel) otherwise.
int decode_gqa_rows(const DecodeCall& c, hipStream_t stream) {
  const int mode = gqa_mode();
  if (mode == 0 || c.heads == c.kv_heads || c.head_dim != 128 || (c.dtype != SPATTEN_BF16 && c.dtype != SPATTEN_F16)) return SPATTEN_ERR_UNSUPPORTED;
  if (c.n_q != 1 || c.mask || c.pq || c.acc || c.prev_scores || c.qkv_x || c.head_ids || c.head_abs || c.flags != 0 || c.causal ||
      !c.q || !c.kr_cache || !c.v_cache || !c.out)
    return SPATTEN_ERR_UNSUPPORTED;
  const int G = c.heads / c.kv_heads;
  if (G > 32 || c.lse_q > 1) return SPATTEN_ERR_UNSUPPORTED;
  // the split layout follows the LAYOUT length (a static launch laid out for a longer cache: the decomposition — and with it
  // every bit of `out` — of the device-length form whose bound is that length); so does the choice of the kernel, or a static
  // launch and the device-length form of the same step could take different kernels around the threshold
  const int lay = (!c.step && c.layout_len > c.kv_len) ? c.layout_len : c.kv_len;
  if ((int64_t)c.kv_len * 256 >= 0x7FFFFFFFll) return SPATTEN_ERR_UNSUPPORTED;          // 32-bit byte offsets inside a plane
  if ((c.kv_sb | c.kv_sh) % 8 != 0 || (c.q_sb | c.q_sh) % 8 != 0 || (c.k_new && (c.new_sb | c.new_sh) % 8 != 0)) return SPATTEN_ERR_UNSUPPORTED;
  const int units = c.batch * c.heads;
  const int ws_splits = c.ws_splits > 0 ? c.ws_splits : kDecodeMaxSplits;
  const int cols = c.batch * c.kv_heads;
  int S, chunk32;
  gqa_layout(cols, lay, c.n_splits, ws_splits, S, chunk32);
  if (mode < 0 && !gqa_pays(c.batch, c.heads, c.kv_heads, lay, chunk32)) return SPATTEN_ERR_UNSUPPORTED;
  if (S > 1 && (!c.workspace || (size_t)units > c.ws_units)) return SPATTEN_ERR_UNSUPPORTED;
  static int env_poll = -1;
  if (env_poll < 0) { const char* e = getenv("SPATTEN_DECODE_POLL"); env_poll = e ? atoi(e) : 1; }
  const int poll_merge = (env_poll != 0 && S > 1 && (long long)S * cols <= coresident_workgroups()) ? 1 : 0;
  const size_t cnt_bytes = decode_cnt_bytes(c.ws_units);

#define SPATTEN_GQA_FILL(T)                                                                                      \
  GqaParams<T> p;                                                                                                \
  p.q = (const T*)c.q; p.q_sb = c.q_sb; p.q_sh = c.q_sh;                                                         \
  p.kc = (T*)c.k_cache; p.krc = (T*)c.kr_cache; p.vc = (T*)c.v_cache; p.kv_sb = c.kv_sb; p.kv_sh = c.kv_sh;      \
  p.k_new = (const T*)c.k_new; p.v_new = (const T*)c.v_new; p.new_sb = c.new_sb; p.new_sh = c.new_sh;            \
  p.append = c.k_new != nullptr;                                                                                 \
  p.cos = (const T*)c.cos; p.sin = (const T*)c.sin; p.table_rows = c.table_rows;                                 \
  p.nr_row = (c.kv_len < c.table_rows ? c.kv_len : c.table_rows) - 1;                                            \
  p.step = (const int32_t*)c.step;                                                                               \
  if (c.step) {   /* the state's staged rotary rows: row 0 = query, row 1 = appended key */                      \
    p.cos = (const T*)((const char*)c.step + kStepHeader);                                                       \
    p.sin = p.cos + 2 * (c.head_dim / 2); p.table_rows = 2; p.nr_row = 1;                                        \
  }                                                                                                              \
  p.pos_ids = c.position_ids; p.pos_sb = c.pos_sb; p.pos_q = c.step ? 0 : c.pos_q;                               \
  p.out = (T*)c.out; p.out_sb = c.out_sb;                                                                        \
  p.scores = (T*)c.scores; p.sc_sb = c.sc_sb; p.sc_sh = c.sc_sh;                                                 \
  p.sc_vec = (c.scores && ((uintptr_t)c.scores % 8 == 0) && c.sc_sb % 4 == 0 && c.sc_sh % 4 == 0) ? 1 : 0;      \
  if (p.sc_vec && (uintptr_t)c.scores % 16 == 0 && c.sc_sb % 8 == 0 && c.sc_sh % 8 == 0 && sizeof(T) == 2) p.sc_vec = 2; \
  p.lse = c.lse;                                                                                                 \
  p.ws_err = (unsigned*)c.workspace;                                                                             \
  p.ws_cnt = c.workspace ? (unsigned*)((char*)c.workspace + kDecodeWsHeader) : nullptr;                          \
  p.ws_part = c.workspace ? (unsigned long long*)((char*)c.workspace + kDecodeWsHeader + cnt_bytes) : nullptr;   \
  p.ws_unit = (int64_t)ws_splits * (c.head_dim + 2);                                                             \
  p.B = c.batch; p.H = c.heads; p.Hkv = c.kv_heads; p.G = G; p.N = c.kv_len; p.S = S; p.chunk = chunk32;         \
  p.poll_merge = poll_merge; p.sqrt_d = sqrtf((float)c.head_dim);                                                \
  const dim3 grid(S, c.kv_heads, c.batch);                                                                       \
  if (c.step) hipLaunchKernelGGL((decode_gqa_kernel<T, true>), grid, dim3(256), 0, stream, p);                   \
  else hipLaunchKernelGGL((decode_gqa_kernel<T, false>), grid, dim3(256), 0, stream, p);                         \
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;

  if (c.dtype == SPATTEN_F16) { SPATTEN_GQA_FILL(f16_t) }
  SPATTEN_GQA_FILL(bf16_t)
#undef SPATTEN_GQA_FILL
}

}  // namespace spatten

#ifdef SPATTEN_GQA_TRACE
extern "C" int spatten_debug_set_gqa_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gqa_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int spatten_decode_gqa_selected(int dtype, int batch, int heads, int kv_heads, int head_dim, int kv_len_layout) {
  if (dtype != SPATTEN_F16 && dtype != SPATTEN_BF16) return 0;
  if (batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || heads == kv_heads || heads / kv_heads > 32) return 0;
  if (head_dim != 128 || kv_len_layout <= 0 || (int64_t)kv_len_layout * 256 >= 0x7FFFFFFFll) return 0;
  const int mode = spatten::gqa_mode();
  if (mode >= 0) return mode;
  int S, chunk32;
  spatten::gqa_layout(batch * kv_heads, kv_len_layout, 0, spatten::kDecodeMaxSplits, S, chunk32);
  return spatten::gqa_pays(batch, heads, kv_heads, kv_len_layout, chunk32) ? 1 : 0;
}

extern "C" int spatten_decode_set_gqa(int mode) {
  if (mode < -1 || mode > 1) return SPATTEN_ERR_INVALID;
  const int prev = spatten::gqa_mode() + 1;
  spatten::g_gqa_mode.store(mode, std::memory_order_relaxed);
  return prev;
}
