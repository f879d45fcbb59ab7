// local_v.hip — the decode step with LOCAL VALUE PRUNING as ONE launch (PARITY UNPINNED: restated from
// the RTL control flow and checked against oracle/spatten_oracle.py:local_value_prune).
//
//   SpAttenController.scala:546-558   after the softmax of a head, the top `val_fetch_num` probabilities of its `key_fetch_num`
//                                     keys are selected (the TopK engine: k-th largest, ties lowest index first, H1) ...
//   SpAttenController.scala:591-612   ... and only those V rows are fetched; P.V runs over them with the probabilities as they
//                                     are (no renormalisation); the stage is skipped when val_fetch_num >= key_fetch_num
//
// r02 / r03 ran this as three dependent chip-wide launches (scores-only decode -> per-head top-k -> gather P.V): 67 us at
// 16384 rows x 40 heads, 30 % kept, against 56 us for the plain fused decode — a slowdown.  Here the
// splits of a head are co-resident BY CONSTRUCTION (grid <= CUs: every workgroup is running or will start without another
// having to finish), so the three phases run inside one launch and hand over through {value, tag} granules (the decode
// kernel's publication protocol, one memory hop per hand-over, placement independent):
//
//   phase 1  every split streams its chunk of the rotated keys ONCE, leaves the logits in the stash (model dtype, the
//            reference's roundings, modify_llama.py:111-119) and keeps them in LDS as order-preserving integer keys; (max, sum)
//            of the chunk go out with the first histogram
//   phase 2  exact k-th largest logit of the HEAD by radix select, 8 bits per pass (2 passes for 16-bit logits, 4 for fp32):
//            every split publishes the 256-bin histogram of its chunk, every split reads all of them — no atomics, nothing to
//            clear, the same sum everywhere — and narrows the prefix; ties at the threshold go to the lowest indices: split s
//            keeps the first (t - ties of the splits before it) of its own
//   phase 3  every split compacts ITS kept rows (order preserved), gathers only those V rows — probabilities with the head's
//            full denominator — and publishes its partial sum; the head's last split adds the partials in split order
//
// Nothing in the launch appends (spatten_kv_append[_step] first).  DYN: the cache length and hence the kept count
// ceil(fraction * length) are read from the device-resident step state, so the step is capturable.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace spatten {

constexpr int kLvThreads = 256;
constexpr int kLvMaxSplits = 32;
constexpr int kLvMaxPasses = 4;
// granules of one split's slot: a 256-bin histogram per pass, 4 auxiliary words, the partial output (D <= 256)
constexpr int kLvSlot = kLvMaxPasses * 256 + 4 + 256;
constexpr size_t kLvHeader = 256;

template <typename T>
struct LvParams {
  const T* q; int64_t q_sb, q_sh;
  const T* krc; const T* vc; int64_t kv_sb, kv_sh;
  const T* cos; const T* sin; int table_rows; int pos_q;
  T* out; int64_t out_sb;
  T* scores; int64_t sc_sb, sc_sh;
  float* lse;
  unsigned long long* ws; unsigned* ws_gen; unsigned* ws_err;
  const int32_t* step;
  double keep_frac; int keep;
  int B, H, Hkv, N, S, chunk;
  float sqrt_d;
  // the step's APPEND inside the launch (round 5; optional): row N - 1 of kc (optional) / krc / vc <- k_new / v_new [B,Hkv,d]
  // (modify_llama.py:95-104), rotated with rotary row `nr_row` of cos / sin (device-length form: the state's staged row 1)
  const T* k_new; const T* v_new; int64_t new_sb, new_sh; T* kc; T* krc_w; T* vc_w; int nr_row;
};

// order-preserving integer key of a logit that IS a model-dtype value (NaN largest, -0 == +0: the order torch.topk ranks by)
template <typename T> struct OKey {       // 16-bit dtypes: the 16-bit pattern
  using type = uint16_t;
  static constexpr int kPasses = 2;
  __device__ static inline uint32_t from(float s) {
    const T v = DT<T>::from_f32(s);
    uint32_t u = *reinterpret_cast<const uint16_t*>(&v);
    if (s != s) return 0xFFFFu;
    if ((u & 0x7FFFu) == 0u) u = 0u;
    return (u & 0x8000u) ? (~u & 0xFFFFu) : (u | 0x8000u);
  }
  __device__ static inline float to(uint32_t k) {
    const uint16_t u = (uint16_t)((k & 0x8000u) ? (k ^ 0x8000u) : (~k & 0xFFFFu));
    return DT<T>::to_f32(*reinterpret_cast<const T*>(&u));
  }
};
template <> struct OKey<float> {
  using type = uint32_t;
  static constexpr int kPasses = 4;
  __device__ static inline uint32_t from(float s) { return ordered_key(s); }
  __device__ static inline float to(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
};

__device__ inline void lv_store(unsigned long long* g, unsigned v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long lv_load(const unsigned long long* g) {
  return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef SPATTEN_LV_TRACE     // developer instrumentation (tools/mb/lv_trace.py): phase timestamps of the workgroups of head 0
__device__ unsigned long long* g_lv_trace = nullptr;
#define LV_STAMP(slot)                                                                                      \
  do {                                                                                                      \
    if (g_lv_trace && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)                               \
      g_lv_trace[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter();                                  \
  } while (0)
#else
#define LV_STAMP(slot)
#endif
#ifndef SPATTEN_LV_UK
#define SPATTEN_LV_UK 6          // key row-groups per pipelined tile of phase 1
#endif
#ifndef SPATTEN_LV_UV
#define SPATTEN_LV_UV 6          // value row-groups per pipelined tile of phase 3
#endif

template <typename T, int D, bool DYN>
__global__ __launch_bounds__(kLvThreads) void local_v_kernel(const LvParams<T> p) {
  constexpr int LPR = D / 16, RPI = kLvThreads / LPR, HALF = D / 2;
  constexpr int UK = SPATTEN_LV_UK, UV = SPATTEN_LV_UV;
  constexpr int NP = OKey<T>::kPasses;
  using V8 = Vec8<T>;
  using raw_t = typename V8::raw;
  using D8 = Dot8<T>;
  using key_t = typename OKey<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_scan[260];
  __shared__ float s_red[4][D + 2];
  __shared__ unsigned s_misc[8];
  __shared__ unsigned s_bef[256], s_mine[256];
  __shared__ float s_aux[2][kLvMaxSplits];

  const int tid = threadIdx.x, c = tid % LPR, r = tid / LPR, wave = tid / kWave, lane = tid % kWave;
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hkv = (p.Hkv == p.H) ? h : h / (p.H / p.Hkv);
  const int unit = b * p.H + h;

  int n_dyn = 0;
  LV_STAMP(0);
  if (DYN) n_dyn = p.step[opaque_lane(0)];
  s_hist[tid] = 0u;               // the first radix pass's histogram is counted WHILE the keys stream (round 5)
  const int lo = split * p.chunk;
  const int rl = min(lo + p.chunk, p.N);                 // static limit of the load addresses (DYN: p.N is the bound)
  key_t* skey = reinterpret_cast<key_t*>(smem);          // [chunk] ordered keys of this split's logits
  uint16_t* klist = reinterpret_cast<uint16_t*>(smem + (size_t)((p.chunk * sizeof(key_t) + 15) / 16) * 16);   // [chunk] kept rows

  const T* krbase = p.krc + b * p.kv_sb + hkv * p.kv_sh;
  const T* vbase = p.vc + b * p.kv_sb + hkv * p.kv_sh;

  // ================================ phase 1: logits of the chunk ====================================================
  struct KTile { raw_t k_lo[UK], k_hi[UK]; };
  KTile ka, kb;
  auto issue_k = [&](KTile& tl, int t0) {
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int j = max(min(t0 + u * RPI + r, rl - 1), 0);
      const T* kp = krbase + (int64_t)j * D;
      tl.k_lo[u] = V8::ldg_stream(kp + 8 * c);
      tl.k_hi[u] = V8::ldg_stream(kp + HALF + 8 * c);
    }
  };
  raw_t q_raw[4];
  {
    const T* qp = p.q + b * p.q_sb + h * p.q_sh;
    const int pq = min(max(p.pos_q, 0), p.table_rows - 1);
    q_raw[0] = V8::ldg(qp + 8 * c);
    q_raw[1] = V8::ldg(qp + HALF + 8 * c);
    q_raw[2] = V8::ldg(p.cos + (int64_t)pq * HALF + 8 * c);
    q_raw[3] = V8::ldg(p.sin + (int64_t)pq * HALF + 8 * c);
  }
  issue_k(ka, lo);
  // APPEND (round 5): every split's first thread-row requests the new token's K / V pieces and the rotary row of its slot (static
  // addresses: nothing waits for the length); the split that owns row N - 1 rotates, stores the three rows in front of the barrier
  // below (whose fence completes the stores: the value gather of phase 3 may read the V row back) and scores the row from
  // registers as one extra key — the key tiles never read it
  const bool app = p.k_new != nullptr;
  raw_t nk_raw[2], nv_raw[2], nr_raw[2];
  if (app && tid < LPR) {
    const T* kp = p.k_new + b * p.new_sb + hkv * p.new_sh;
    const T* vp = p.v_new + b * p.new_sb + hkv * p.new_sh;
    nk_raw[0] = V8::ldg(kp + 8 * c); nk_raw[1] = V8::ldg(kp + HALF + 8 * c);
    nv_raw[0] = V8::ldg(vp + 8 * c); nv_raw[1] = V8::ldg(vp + HALF + 8 * c);
    nr_raw[0] = V8::ldg(p.cos + (int64_t)p.nr_row * HALF + 8 * c);
    nr_raw[1] = V8::ldg(p.sin + (int64_t)p.nr_row * HALF + 8 * c);
  }
  const unsigned gen = p.ws_gen[unit + opaque_lane(0)];
  __builtin_amdgcn_sched_barrier(0);
  const int N = DYN ? __builtin_amdgcn_readfirstlane(n_dyn) : p.N;
  const int hi = min(lo + p.chunk, N);
  // (DYN: p.N is the launch's BOUND = the planes' capacity — a replay past it must not write beyond the planes: ADVICE r05)
  const bool owns_new = app && lo <= N - 1 && N - 1 < hi && (!DYN || N <= p.N);       // (wave-uniform)
  const int hi_t = owns_new ? hi - 1 : hi;                       // rows the key tiles score
  typename D8::packed nk_lo, nk_hi;                              // the appended key, rotated (owner's first thread-row)
  if (owns_new && tid < LPR) {
    float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
    V8::unpack(nk_raw[0], xlo);
    V8::unpack(nk_raw[1], xhi);
    V8::unpack(nr_raw[0], cc);
    V8::unpack(nr_raw[1], ss);
    rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
    nk_lo = D8::pack(ylo);
    nk_hi = D8::pack(yhi);
    const int64_t dst = b * p.kv_sb + hkv * p.kv_sh + (int64_t)(N - 1) * D;
    if (p.kc) { V8::stg(p.kc + dst + 8 * c, nk_raw[0]); V8::stg(p.kc + dst + HALF + 8 * c, nk_raw[1]); }
    V8::stg(p.krc_w + dst + 8 * c, V8::pack(ylo));
    V8::stg(p.krc_w + dst + HALF + 8 * c, V8::pack(yhi));
    V8::stg(p.vc_w + dst + 8 * c, nv_raw[0]);
    V8::stg(p.vc_w + dst + HALF + 8 * c, nv_raw[1]);
  }
  __syncthreads();                 // (the zeroed histogram, before any wave's first count; the tile loads are in flight;
                                   //  the appended rows are in memory)
  const int n_loc = max(hi - lo, 0);
  int keep = p.keep;
  if (DYN) keep = (int)ceil(p.keep_frac * (double)N);
  keep = max(1, min(keep, N));

  typename D8::packed q_lo, q_hi;
  {
    float xlo[8], xhi[8], cc[8], ss[8], ylo[8], yhi[8];
    V8::unpack(q_raw[0], xlo);
    V8::unpack(q_raw[1], xhi);
    V8::unpack(q_raw[2], cc);
    V8::unpack(q_raw[3], ss);
    rope_pair<T>(xlo, xhi, cc, ss, ylo, yhi);
    q_lo = D8::pack(ylo);
    q_hi = D8::pack(yhi);
  }
  const float rsqrt_d = 1.0f / p.sqrt_d;
  T* stashp = p.scores + b * p.sc_sb + h * p.sc_sh;
  float m_run = -INFINITY, l_run = 0.f;
  auto score_tile = [&](KTile& tl, int t0) {
    float sc[UK];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const float a = group_sum<LPR>(D8::dot(q_hi, tl.k_hi[u], D8::dot(q_lo, tl.k_lo[u], 0.f)));
      sc[u] = DT<T>::round(div_by_const(DT<T>::round(a), p.sqrt_d, rsqrt_d));        // modify_llama.py:111-113
    }
    float m_new = m_run;
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int j = t0 + u * RPI + r;
      const bool valid = j < hi_t;
      if (c == 0 && valid) {
        stashp[j] = DT<T>::from_f32(sc[u]);                                          // :116-119
        const uint32_t ok = OKey<T>::from(sc[u]);
        skey[j - lo] = (key_t)ok;
        atomicAdd(&s_hist[(ok >> (8 * (NP - 1))) & 255u], 1u);                       // most significant digit (phase 2, pass 0)
      }
      sc[u] = valid ? sc[u] : -INFINITY;
      m_new = fmaxf(m_new, sc[u]);
    }
    if (m_new > m_run) { l_run *= __expf(m_run - m_new); m_run = m_new; }
#pragma unroll
    for (int u = 0; u < UK; ++u) l_run += (sc[u] == -INFINITY) ? 0.f : __expf(sc[u] - m_run);
  };
  if (owns_new && tid < LPR) {      // the appended key: one extra row of the first thread-row's softmax (same arithmetic as a tile row)
    const float a = group_sum<LPR>(D8::dot(q_hi, nk_hi, D8::dot(q_lo, nk_lo, 0.f)));
    const float sn = DT<T>::round(div_by_const(DT<T>::round(a), p.sqrt_d, rsqrt_d));
    if (c == 0) {
      stashp[N - 1] = DT<T>::from_f32(sn);
      const uint32_t ok = OKey<T>::from(sn);
      skey[N - 1 - lo] = (key_t)ok;
      atomicAdd(&s_hist[(ok >> (8 * (NP - 1))) & 255u], 1u);
    }
    m_run = sn; l_run = 1.f;
  }
  constexpr int KT = RPI * UK;
  for (int t0 = lo; t0 < hi_t; t0 += 2 * KT) {
    issue_k(kb, t0 + KT);
    __builtin_amdgcn_sched_barrier(0);
    score_tile(ka, t0);
    issue_k(ka, t0 + 2 * KT);
    __builtin_amdgcn_sched_barrier(0);
    if (t0 + KT < hi_t) score_tile(kb, t0 + KT);
  }
  LV_STAMP(1);      // end of the key stream
  // (max, sum) of the chunk: every row's LPR lanes agree, so only the row's first lane contributes its sum
  float m_s, l_s;
  {
    const float mw = wave_max(m_run);
    const float lw = wave_sum((c == 0 && m_run != -INFINITY) ? l_run * __expf(m_run - mw) : 0.f);
    if (lane == 0) { s_red[wave][0] = mw; s_red[wave][1] = lw; }
    __syncthreads();                                     // also: every key of the chunk is in LDS
    const float m0 = s_red[0][0], m1 = s_red[1][0], m2 = s_red[2][0], m3 = s_red[3][0];
    m_s = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float mu = (m_s == -INFINITY) ? 0.f : m_s;
    l_s = (s_red[0][1] * __expf(m0 - mu) + s_red[1][1] * __expf(m1 - mu)) + (s_red[2][1] * __expf(m2 - mu) + s_red[3][1] * __expf(m3 - mu));
  }

  // ================================ phase 2: the head's k-th largest logit ===========================================
  unsigned long long* wsu = p.ws + (int64_t)unit * (kLvMaxSplits * kLvSlot);
  unsigned long long* slot = wsu + (int64_t)split * kLvSlot;
  const unsigned tag = (gen & 0x7FFFFFFFu) + 1u;
  bool expired = false;
  uint32_t prefix = 0;            // the digits of the threshold found so far (most significant first)
  int need = keep;                // rank of the threshold among the keys that match the prefix (1 = the largest)
  unsigned ties_before = 0, ties_mine = 0;
  float m_g = m_s, l_g = l_s;
  // bin `tid` of one histogram of every split: 8 loads per round trip; adds up all splits (tot) and the splits before this one
  // (with_aux: threads 0 .. S-1 fetch their split's (max, sum) pair in the SAME batch — a separate poll was a second ~2 us
  //  round trip through the memory side in front of the second pass, phase stamps r05)
  auto gather_bins = [&](auto region_of, unsigned want_tag, unsigned& tot, unsigned& before, bool with_aux) {
    tot = 0; before = 0;
    for (int s0 = 0; s0 < p.S; s0 += 8) {
      unsigned long long g[8], a0 = 0, a1 = 0;
      const bool aux_here = with_aux && s0 == 0 && tid < p.S;
      const unsigned long long* ax = wsu + (int64_t)min(tid, p.S - 1) * kLvSlot + kLvMaxPasses * 256;
      int spins = 0;
      bool landed;
      do {
        unsigned diff = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int s = min(s0 + k, p.S - 1);
          g[k] = lv_load(wsu + (int64_t)s * kLvSlot + region_of(s) * 256 + tid);
          diff |= (unsigned)(g[k] >> 32) ^ want_tag;
        }
        if (aux_here) {
          a0 = lv_load(ax); a1 = lv_load(ax + 1);
          diff |= ((unsigned)(a0 >> 32) ^ want_tag) | ((unsigned)(a1 >> 32) ^ want_tag);
        }
        landed = diff == 0u;
      } while (!landed && ++spins < (1 << 16));
      expired |= !landed;
      if (aux_here) { s_aux[0][tid] = __uint_as_float((unsigned)a0); s_aux[1][tid] = __uint_as_float((unsigned)a1); }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s0 + k < p.S) { tot += (unsigned)g[k]; if (s0 + k < split) before += (unsigned)g[k]; }
    }
  };
  // the bin that holds the `need`-th largest key: suffix sums over the 256 bins (keys are ranked largest first), exactly one bin
  // brackets the rank.  Round 5: ONE barrier — every wave scans all 256 totals itself (4 descending bins per lane, one wave scan,
  // a ballot finds the bracketing bin) and reads `before` / `mine` of that bin from LDS (r04: wave 0 scanned, the pick travelled
  // through s_misc: three barriers, ~1.9k cycles per pick by the phase stamps).  Returns {bin, keys above it, before, mine}.
  struct Pick { unsigned bin, above, before, mine; };
  auto pick_bin = [&](unsigned tot, unsigned before, unsigned mine, int rank) -> Pick {
    s_scan[tid] = tot;
    s_bef[tid] = before;
    s_mine[tid] = mine;
    __syncthreads();
    unsigned v4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = s_scan[255 - (4 * lane + i)];       // descending bins
    const unsigned mysum = v4[0] + v4[1] + v4[2] + v4[3];
    unsigned incl = mysum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    unsigned run = incl - mysum, hb = 0, ha = 0;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (run < (unsigned)rank && (unsigned)rank <= run + v4[i]) { hit = true; hb = 255u - (unsigned)(4 * lane + i); ha = run; }
      run += v4[i];
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    const int L = m ? (int)__builtin_ctzll(m) : 0;
    Pick pk;
    pk.bin = (unsigned)__shfl((int)hb, L, kWave);
    pk.above = (unsigned)__shfl((int)ha, L, kWave);
    pk.before = s_bef[pk.bin];
    pk.mine = s_mine[pk.bin];
    return pk;
  };
  // keys of this split whose digits above `shift + 8` equal `pfx`, counted by their digit at `shift` into hist[256]
  // (16-byte LDS reads: 8 / 4 keys per lane and read; r04 read them one by one: 3.2 us of a pass at 2731 keys per split)
  auto count_digit = [&](unsigned* hist, uint32_t pfx, int shift) {
    constexpr int KPV = 16 / (int)sizeof(key_t);
    for (int i0 = tid * KPV; i0 < n_loc; i0 += kLvThreads * KPV) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(skey + i0);
      const key_t* kv = reinterpret_cast<const key_t*>(&v);
#pragma unroll
      for (int j = 0; j < KPV; ++j) {
        const uint32_t k = kv[j];
        if (i0 + j < n_loc && (k >> (shift + 8)) == pfx) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
    }
  };
  // (r05, measured and dropped: for 16-bit logits ONE publish round — the low-byte histograms of three candidate high bytes
  //  published beside the first histogram, then all four regions + the auxiliary words fetched in one batch of loads: a round trip
  //  through the memory side is ~2 us under this load and grows with what it carries; phase 2 stayed at 9-10 us.  HISTORY.md)
  // the head's (max, sum): every split's pair (published with its first histogram), folded in split order by every thread
  auto fold_max_sum = [&]() {       // (s_aux was filled by the first gather; pick_bin's barriers stand between)
    float mm = -INFINITY;
    for (int s = 0; s < p.S; ++s) mm = fmaxf(mm, s_aux[0][s]);
    const float mu = (mm == -INFINITY) ? 0.f : mm;
    float ll = 0.f;
    for (int s = 0; s < p.S; ++s) ll += s_aux[1][s] * __expf(s_aux[0][s] - mu);
    m_g = mm; l_g = ll;
  };
  auto take_pick = [&](int bits_done, const Pick& pk) {
    prefix = (bits_done == 0) ? pk.bin : ((prefix << 8) | pk.bin);
    need -= (int)pk.above;
    ties_before = pk.before;
    ties_mine = pk.mine;
  };
  __syncthreads();                                  // the stream's counts of the most significant digit are complete
  LV_STAMP(2);
  if (p.S == 1) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      if (pass > 0) {
        s_hist[tid] = 0u;
        __syncthreads();
        count_digit(s_hist, prefix, 8 * (NP - 1 - pass));
        __syncthreads();
      }
      const Pick pk = pick_bin(s_hist[tid], 0u, s_hist[tid], need);
      take_pick(pass, pk);
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      if (pass > 0) {
        __syncthreads();
        s_hist[tid] = 0u;
        __syncthreads();
        count_digit(s_hist, prefix, 8 * (NP - 1 - pass));
        __syncthreads();
        if (pass == 1) LV_STAMP(5);     // second histogram built
      }
      lv_store(slot + pass * 256 + tid, s_hist[tid], tag);
      if (pass == 0 && tid < 2) lv_store(slot + kLvMaxPasses * 256 + tid, __float_as_uint(tid == 0 ? m_s : l_s), tag);
      unsigned tot, before;
      gather_bins([&](int) { return pass; }, tag, tot, before, pass == 0);
      if (pass == 0) LV_STAMP(3);       // first histograms of every split read
      if (pass == 1) LV_STAMP(6);       // second histograms read
      const Pick pk = pick_bin(tot, before, s_hist[tid], need);
      take_pick(pass, pk);
      if (pass == 0) fold_max_sum();
    }
  }
  // threshold key = prefix; `need` of the keys EQUAL to it are kept, lowest index first: this split takes what the
  // splits before it leave
  LV_STAMP(8);        // threshold known
  const uint32_t thr = prefix;
  const int t_mine = max(0, min((int)ties_mine, need - (int)ties_before));

  // ================================ phase 3: compact the kept rows, gather their V rows ============================
  // order-preserving compaction (round 5): a thread owns a CONTIGUOUS run of keys (16-byte LDS reads), counts its (greater, equal)
  // keys, one exclusive scan over the 256 threads' packed counts, then every thread writes its kept rows at its offset — one pass
  // over the keys each way (r04: two passes of 64-key segments with ballots per segment: 4.4 us at 2731 keys, phase stamps)
  constexpr int KPV = 16 / (int)sizeof(key_t);
  const int run = ((n_loc + kLvThreads - 1) / kLvThreads + KPV - 1) / KPV * KPV;      // keys per thread, whole 16-byte reads
  const int r0 = tid * run;
  unsigned cnt_gt = 0, cnt_eq = 0;
  for (int i0 = r0; i0 < r0 + run && i0 < n_loc; i0 += KPV) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(skey + i0);
    const key_t* kv = reinterpret_cast<const key_t*>(&v);
#pragma unroll
    for (int j = 0; j < KPV; ++j) {
      const uint32_t k = kv[j];
      const bool in = i0 + j < n_loc;
      cnt_gt += (in && k > thr) ? 1u : 0u;
      cnt_eq += (in && k == thr) ? 1u : 0u;
    }
  }
  unsigned* seg = s_scan;                                // [256] exclusive prefix of the packed (gt << 16 | eq) counts; [256] the total
  {
    const unsigned mine = (cnt_gt << 16) | cnt_eq;       // (chunk <= 16384 rows: either count fits 15 bits)
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_misc[4 + wave] = incl;
    __syncthreads();
    unsigned base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += (w < wave) ? s_misc[4 + w] : 0u;
    seg[tid] = base + incl - mine;
    if (tid == kLvThreads - 1) seg[256] = base + incl;
  }
  __syncthreads();
  {
    const unsigned ex = seg[tid];
    unsigned g_run = ex >> 16, e_run = ex & 0xFFFFu;
    for (int i0 = r0; i0 < r0 + run && i0 < n_loc; i0 += KPV) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(skey + i0);
      const key_t* kv = reinterpret_cast<const key_t*>(&v);
#pragma unroll
      for (int j = 0; j < KPV; ++j) {
        const uint32_t k = kv[j];
        const bool in = i0 + j < n_loc;
        const bool gt = in && k > thr, eq = in && k == thr;
        // kept rows before this one = greater rows before + min(equal rows before, t_mine)
        if (gt || (eq && (int)e_run < t_mine)) klist[g_run + min(e_run, (unsigned)t_mine)] = (uint16_t)(i0 + j);
        g_run += gt ? 1u : 0u;
        e_run += eq ? 1u : 0u;
      }
    }
  }
  __syncthreads();
  const int n_kept = (int)(seg[256] >> 16) + t_mine;     // greater rows of the split + its share of the ties
  LV_STAMP(9);        // kept rows compacted
  const float mu_g = (m_g == -INFINITY) ? 0.f : m_g;
  const float rl_g = 1.0f / l_g;
  struct VTile { raw_t v_lo[UV], v_hi[UV]; float pj[UV]; };
  VTile va, vb;
  auto issue_v = [&](VTile& tl, int g0) {
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int i = (g0 + u) * RPI + r;
      const int li = (int)klist[min(i, max(n_kept - 1, 0))];
      const T* vp = vbase + (int64_t)(lo + li) * D;
      tl.v_lo[u] = V8::ldg_stream(vp + 8 * c);
      tl.v_hi[u] = V8::ldg_stream(vp + HALF + 8 * c);
      // (computed for every lane — li is always a valid entry — and selected: under the condition the compiler branches around the
      //  LDS read + exponential, between the value loads, and the loop's vmcnt waits degrade to a full drain per iteration)
      const float pe = __expf(OKey<T>::to((uint32_t)skey[li]) - mu_g) * rl_g;
      tl.pj[u] = (i < n_kept) ? pe : 0.f;
    }
  };
  float olo[8], ohi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = 0.f; ohi[i] = 0.f; }
  auto pv_tile = [&](VTile& tl) {
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      float a[8], bq[8];
      V8::unpack(tl.v_lo[u], a);
      V8::unpack(tl.v_hi[u], bq);
#pragma unroll
      for (int e = 0; e < 8; ++e) { olo[e] = fmaf(tl.pj[u], a[e], olo[e]); ohi[e] = fmaf(tl.pj[u], bq[e], ohi[e]); }
    }
  };
  const int n_grp = (n_kept + RPI - 1) / RPI;
  if (n_kept > 0) {
    issue_v(va, 0);
    for (int g0 = 0; g0 < n_grp; g0 += 2 * UV) {
      issue_v(vb, g0 + UV);
      __builtin_amdgcn_sched_barrier(0);
      pv_tile(va);
      issue_v(va, g0 + 2 * UV);
      __builtin_amdgcn_sched_barrier(0);
      if (g0 + UV < n_grp) pv_tile(vb);
    }
  }
  LV_STAMP(10);       // V rows gathered
  // lanes with equal c across the wave's row groups, then the 4 waves (decode_attn.hip's reduction)
  if (LPR == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      olo[i] += dpp_mov<kDppRor8>(olo[i]); olo[i] += dpp_mov<kDppRor4>(olo[i]);
      ohi[i] += dpp_mov<kDppRor8>(ohi[i]); ohi[i] += dpp_mov<kDppRor4>(ohi[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) { olo[i] += dpp_mov<kDppRor8>(olo[i]); ohi[i] += dpp_mov<kDppRor8>(ohi[i]); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { olo[i] = xor32_sum(xor16_sum(olo[i])); ohi[i] = xor32_sum(xor16_sum(ohi[i])); }
  __syncthreads();
  if (lane < LPR) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_red[wave][8 * lane + i] = olo[i]; s_red[wave][HALF + 8 * lane + i] = ohi[i]; }
  }
  __syncthreads();
  float o_tot = 0.f;
  if (tid < D) o_tot = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
  T* outp = p.out + b * p.out_sb + (int64_t)h * D;
  if (p.S == 1) {
    if (tid < D) outp[tid] = DT<T>::from_f32(o_tot);
    if (p.lse != nullptr && tid == 0) { p.lse[unit * 2] = m_g; p.lse[unit * 2 + 1] = l_g; }
    if (tid == 0) p.ws_gen[unit] = gen + 1u;
    return;
  }
  if (tid < D) lv_store(slot + kLvMaxPasses * 256 + 4 + tid, __float_as_uint(o_tot), tag);
  LV_STAMP(11);       // partial published
  if (split != p.S - 1) return;
  float og = 0.f;
  if (tid < D) {
    for (int s0 = 0; s0 < p.S; s0 += 8) {
      unsigned long long g[8];
      int spins = 0;
      bool landed;
      do {
        unsigned diff = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int s = min(s0 + k, p.S - 1);
          g[k] = lv_load(wsu + (int64_t)s * kLvSlot + kLvMaxPasses * 256 + 4 + tid);
          diff |= (unsigned)(g[k] >> 32) ^ tag;
        }
        landed = diff == 0u;
      } while (!landed && ++spins < (1 << 16));
      expired |= !landed;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s0 + k < p.S) og += __uint_as_float((unsigned)g[k]);        // split order: deterministic
    }
  }
  if (expired) { atomicOr(p.ws_err, 1u); og = __builtin_nanf(""); }
  LV_STAMP(12);       // merged
  if (tid < D) outp[tid] = DT<T>::from_f32(og);
  if (tid == 0) {
    if (p.lse != nullptr) { p.lse[unit * 2] = m_g; p.lse[unit * 2 + 1] = l_g; }
    p.ws_gen[unit] = gen + 1u;
  }
}

#ifdef SPATTEN_LV_TRACE
}  // namespace spatten
extern "C" int spatten_debug_set_lv_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(spatten::g_lv_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
namespace spatten {
#endif
static inline size_t lv_gen_bytes(size_t units) { return (units * sizeof(unsigned) + 255) / 256 * 256; }

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_local_v_workspace_bytes(int batch, int heads) {
  if (batch <= 0 || heads <= 0) return 0;
  const size_t units = (size_t)batch * heads;
  return kLvHeader + lv_gen_bytes(units) + units * kLvMaxSplits * kLvSlot * sizeof(unsigned long long);
}

static int local_v_launch(int dtype, const void* q, int64_t q_sb, int64_t q_sh, const void* kr_cache,
                          const void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos, const void* sin,
                          int table_rows, int pos_q, void* out, int64_t out_sb, void* scores, int64_t sc_sb,
                          int64_t sc_sh, float* lse, void* workspace, int batch, int heads, int kv_heads,
                          int head_dim, int kv_len, int keep, double keep_fraction, int kv_len_layout,
                          const void* step_state, void* stream, const void* k_new, const void* v_new, int64_t new_sb,
                          int64_t new_sh, void* k_cache) {
  if ((k_new == nullptr) != (v_new == nullptr)) return SPATTEN_ERR_INVALID;
  if (!q || !kr_cache || !v_cache || !cos || !sin || !out || !scores || !workspace) return SPATTEN_ERR_INVALID;
  if (!ok_dtype(dtype) || batch <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads != 0 || kv_len <= 0) return SPATTEN_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  if (step_state ? !(keep_fraction > 0.0 && keep_fraction <= 1.0) : keep <= 0) return SPATTEN_ERR_INVALID;
  if (!step_state && (pos_q < 0 || pos_q >= table_rows)) return SPATTEN_ERR_INVALID;
  const int lay = (!step_state && kv_len_layout > kv_len) ? kv_len_layout : kv_len;
  const long long units = (long long)batch * heads;
  // the splits of a head poll each other: a MULTI-split grid must be resident at once.  More units than CUs run with one split per
  // head — that kernel path polls nobody (refusing it sent B*H > CU-count launches to the slower three-launch form
  // and made the graph step raise)
  const int resident = coresident_workgroups();
  int S = (int)std::max(1LL, resident / units);
  if (S > kLvMaxSplits) S = kLvMaxSplits;
  const int min_rows = 256;                          // a split shorter than this is all latency
  if (S > std::max(1, lay / min_rows)) S = std::max(1, lay / min_rows);
  const int chunk = ceil_div(ceil_div(lay, S), 8) * 8;
  S = ceil_div(lay, chunk);
  if (S > 1 && (long long)S * units > resident) return SPATTEN_ERR_UNSUPPORTED;   // (cascade.local_v_decode: the three-launch path)
  const size_t key_b = dtype == SPATTEN_F32 ? 4 : 2;
  const size_t lds = ((size_t)chunk * key_b + 15) / 16 * 16 + (size_t)chunk * 2;
  if (chunk > 16384 || lds > 120 * 1024) return SPATTEN_ERR_UNSUPPORTED;   // (the three-launch path covers longer chunks)
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)S, (unsigned)heads, (unsigned)batch);
#define SPATTEN_LV(T, DD, DYN_)                                                                                        \
  {                                                                                                                    \
    LvParams<T> p;                                                                                                     \
    p.q = (const T*)q; p.q_sb = q_sb; p.q_sh = q_sh;                                                                   \
    p.krc = (const T*)kr_cache; p.vc = (const T*)v_cache; p.kv_sb = kv_sb; p.kv_sh = kv_sh;                             \
    p.cos = (const T*)cos; p.sin = (const T*)sin; p.table_rows = table_rows; p.pos_q = pos_q;                           \
    p.step = (const int32_t*)step_state;                                                                               \
    if (step_state) {                                                                                                  \
      p.cos = (const T*)((const char*)step_state + kStepHeader); p.sin = p.cos + 2 * (head_dim / 2);                    \
      p.table_rows = 2; p.pos_q = 0;                                                                                   \
    }                                                                                                                  \
    p.out = (T*)out; p.out_sb = out_sb; p.scores = (T*)scores; p.sc_sb = sc_sb; p.sc_sh = sc_sh; p.lse = lse;           \
    p.ws_err = (unsigned*)workspace; p.ws_gen = (unsigned*)((char*)workspace + kLvHeader);                              \
    p.ws = (unsigned long long*)((char*)workspace + kLvHeader + lv_gen_bytes((size_t)units));                           \
    p.keep_frac = keep_fraction; p.keep = keep;                                                                        \
    p.B = batch; p.H = heads; p.Hkv = kv_heads; p.N = kv_len; p.S = S; p.chunk = chunk;                                 \
    p.sqrt_d = sqrtf((float)head_dim);                                                                                 \
    p.k_new = (const T*)k_new; p.v_new = (const T*)v_new; p.new_sb = new_sb; p.new_sh = new_sh; p.kc = (T*)k_cache;     \
    p.krc_w = (T*)const_cast<void*>(kr_cache); p.vc_w = (T*)const_cast<void*>(v_cache);                                 \
    p.nr_row = step_state ? 1 : std::min(kv_len - 1, table_rows - 1);                                                  \
    static bool attr_set[64] = {};       /* the attribute is per DEVICE */                                \
    int dev_ = 0;                                                                                                      \
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) dev_ = 0;                                         \
    if (!attr_set[dev_]) {                                                                                             \
      if (hipFuncSetAttribute((const void*)local_v_kernel<T, DD, DYN_>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess) { \
        (void)hipGetLastError();                                                                                       \
        return SPATTEN_ERR_UNSUPPORTED;                                                                                \
      }                                                                                                                \
      attr_set[dev_] = true;                                                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL((local_v_kernel<T, DD, DYN_>), grid, dim3(kLvThreads), lds, st, p);                              \
  }
#define SPATTEN_LV_D(T, DYN_) { if (head_dim == 128) SPATTEN_LV(T, 128, DYN_) else SPATTEN_LV(T, 64, DYN_) }
#define SPATTEN_LV_T(DYN_)                                                                                             \
  { if (dtype == SPATTEN_BF16) SPATTEN_LV_D(bf16_t, DYN_) else if (dtype == SPATTEN_F16) SPATTEN_LV_D(f16_t, DYN_) else SPATTEN_LV_D(float, DYN_) }
  if (step_state) SPATTEN_LV_T(true) else SPATTEN_LV_T(false)
#undef SPATTEN_LV_T
#undef SPATTEN_LV_D
#undef SPATTEN_LV
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

extern "C" int spatten_attn_decode_local_v(int dtype, const void* q, int64_t q_sb, int64_t q_sh, const void* kr_cache,
                                           const void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos, const void* sin,
                                           int table_rows, int pos_q, void* out, int64_t out_sb, void* scores, int64_t sc_sb,
                                           int64_t sc_sh, float* lse, void* workspace, int batch, int heads, int kv_heads,
                                           int head_dim, int kv_len, int keep, double keep_fraction, int kv_len_layout,
                                           const void* step_state, void* stream) {
  return local_v_launch(dtype, q, q_sb, q_sh, kr_cache, v_cache, kv_sb, kv_sh, cos, sin, table_rows, pos_q, out, out_sb, scores, sc_sb,
                        sc_sh, lse, workspace, batch, heads, kv_heads, head_dim, kv_len, keep, keep_fraction, kv_len_layout, step_state,
                        stream, nullptr, nullptr, 0, 0, nullptr);
}

extern "C" int spatten_attn_decode_local_v_append(int dtype, const void* q, int64_t q_sb, int64_t q_sh, const void* k_new,
                                                  const void* v_new, int64_t new_sb, int64_t new_sh, void* k_cache, void* kr_cache,
                                                  void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos, const void* sin,
                                                  int table_rows, int pos_q, void* out, int64_t out_sb, void* scores,
                                                  int64_t sc_sb, int64_t sc_sh, float* lse, void* workspace, int batch, int heads,
                                                  int kv_heads, int head_dim, int kv_len, int keep, double keep_fraction,
                                                  int kv_len_layout, const void* step_state, void* stream) {
  if (!k_new || !v_new) return SPATTEN_ERR_INVALID;
  return local_v_launch(dtype, q, q_sb, q_sh, kr_cache, v_cache, kv_sb, kv_sh, cos, sin, table_rows, pos_q, out, out_sb, scores, sc_sb,
                        sc_sh, lse, workspace, batch, heads, kv_heads, head_dim, kv_len, keep, keep_fraction, kv_len_layout, step_state,
                        stream, k_new, v_new, new_sb, new_sh, k_cache);
}
