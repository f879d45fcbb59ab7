// decode_chain.hip — the attention path of ALL layers of one decode token in ONE launch (round 6).
//
// Why.  At B = 1 a layer's decode launch (decode_attn.hip) is latency-, not byte-bound: 34 MB of K/V take ~5.4 us at the
// achievable HBM rate, the launch takes ~10.8 — kernel boundary, first-byte latency and the ramp of the stream in front,
// reduce / publish / poll / merge with an idle memory pipe behind.  A layer's K/V tile depends on nothing upstream (the cache
// is known); only its query and appended row do (q_l is a function of out_{l-1}: modify_llama.py:72-92).  So the launch
// boundary can move under the previous layer's tail:
//   grid = (S splits, heads, B x depth); a workgroup WALKS the layers (lane p of `depth` serves layers p, p + depth, ...).  At
//   the top of a layer's step waves 1.. request their rows of the tile at once; wave 0 waits for layer l-1's completion words
//   (one per (b, head), stored by that unit's merger behind its `out` row), stages q / k_new / v_new in LDS, then requests its
//   rows; the step's arithmetic, reduction, publication and merge are decode_body's, bit for bit what the per-layer launch
//   computes.
// The loads are the real ones, issued once, by resident workgroups (not a prefetch touch, not a side stream).
// Per-layer state lives in a DEVICE table (spatten_chain_layer_t), so a captured graph needs no kernel-argument patching.
// What was measured on the way (tools/mb/chain_trace.py, HISTORY.md round 6): a CU's vector-memory path is ONE in-order queue
// shared by its waves — a poll / query fetch / partial read issued behind a tile's 160 requests waits until they have drained
// (6.4 us when the stream is HBM-bound) — and takes ~40 cycles per 1-KB wave load, so (a) a second register tile requested in one
// burst in front of the arithmetic, (b) a tile refilled register group by register group inside the arithmetic and (c) two
// workgroups per CU at 128 registers (spills) were all slower than this form.
//
// Protocol words (chain workspace): header word 0 = error flags (bit 0: a merge timed out, bit 1: a layer wait timed out),
// word 1 = token epoch, word 2 = completed units of the last launched layer; then [layers][fs] completion words; then one
// split-N workspace per layer.  A completion word holds epoch + 1 of the token that wrote it, so nothing is cleared between
// tokens; the unit that completes the last layer advances the epoch.
#include <string.h>

#include "decode_body.h"

namespace spatten {

// The device-side view of spatten_chain_layer_t: the same 80 bytes with the pointers typed as GLOBAL-address-space pointers — a
// pointer loaded from memory is otherwise "generic" and every K/V load through it a flat_load (counted by lgkmcnt as well as
// vmcnt, 64-bit address registers per load instead of a scalar base + 32-bit offset)
#define SPATTEN_GLOBAL __attribute__((address_space(1)))
struct ChainLayerDev {
  SPATTEN_GLOBAL char* k_cache; SPATTEN_GLOBAL char* kr_cache; SPATTEN_GLOBAL char* v_cache;
  const SPATTEN_GLOBAL char* q; const SPATTEN_GLOBAL char* k_new; const SPATTEN_GLOBAL char* v_new;
  SPATTEN_GLOBAL char* out; SPATTEN_GLOBAL char* scores;
  const SPATTEN_GLOBAL int32_t* head_ids; int32_t n_active; int32_t pad_;
};
static_assert(sizeof(ChainLayerDev) == sizeof(spatten_chain_layer_t), "device view of the layer table");

template <typename T>
struct ChainConst {
  DecodeParams<T> base;                 // everything that does not change from layer to layer
  int n_layers, depth, append;
  unsigned* hdr;                        // chain workspace header
  unsigned* flags; int fs;              // completion words [n_layers][fs]
  char* ws_layers; int64_t ws_layer_bytes, cnt_bytes;   // the per-layer split-N workspaces
};

// The layer table, resolved for this workgroup column and staged in LDS ONCE per launch: a step then starts with a few LDS
// reads instead of a chain of dependent scalar loads from memory (table entry -> head list -> head id, the scan for the layer
// it depends on: ~1.5 us in front of every layer's first tile request, measured).  Pointers are kept as 64-bit integers.
constexpr int kChainMaxLayers = 128;
struct ChainLds {
  uint64_t kc, krc, vc, q, kn, vn, out, scores;
  int32_t h, n_active, wait_l, wait_n, last, pad_[3];
};
static_assert(sizeof(ChainLds) == 96, "six 16-byte LDS reads per step");

__device__ __forceinline__ void chain_stage_table(const ChainLayerDev* __restrict__ table, int n_layers, ChainLds* s_tab) {
  for (int l = (int)threadIdx.x; l < n_layers; l += (int)blockDim.x) {
    const ChainLayerDev e = table[l];
    ChainLds t;
    t.kc = (uint64_t)e.k_cache; t.krc = (uint64_t)e.kr_cache; t.vc = (uint64_t)e.v_cache;
    t.q = (uint64_t)e.q; t.kn = (uint64_t)e.k_new; t.vn = (uint64_t)e.v_new;
    t.out = (uint64_t)e.out; t.scores = (uint64_t)e.scores;
    t.n_active = e.n_active;
    t.h = ((int)blockIdx.y < e.n_active && e.head_ids) ? e.head_ids[blockIdx.y] : (int)blockIdx.y;
    t.wait_l = -1; t.wait_n = 0; t.last = 0; t.pad_[0] = t.pad_[1] = t.pad_[2] = 0;
    s_tab[l] = t;
  }
  __syncthreads();
  // the layer this one depends on / whether a later layer launches anything (head pruning can empty a rank's layer)
  for (int l = (int)threadIdx.x; l < n_layers; l += (int)blockDim.x) {
    int lp = l - 1;
    while (lp >= 0 && s_tab[lp].n_active <= 0) --lp;
    bool last = true;
    for (int j = l + 1; j < n_layers && last; ++j) last = s_tab[j].n_active <= 0;
    s_tab[l].wait_l = lp;
    s_tab[l].wait_n = lp >= 0 ? s_tab[lp].n_active : 0;
    s_tab[l].last = last ? 1 : 0;
  }
  __syncthreads();
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

template <typename T>
__device__ __forceinline__ int chain_fill(DecodeParams<T>& p, const ChainConst<T>& c, const ChainLds* s_tab, int l, unsigned tag) {
  const ChainLds e = s_tab[l];                 // (the same address in every lane: a broadcast read)
  const int n_active = __builtin_amdgcn_readfirstlane(e.n_active);
  const int wait_l = __builtin_amdgcn_readfirstlane(e.wait_l), wait_n = __builtin_amdgcn_readfirstlane(e.wait_n);
  const bool last = __builtin_amdgcn_readfirstlane(e.last) != 0;
  const uint64_t qp = uniform_u64(e.q);
  p.kc = (T*)(SPATTEN_GLOBAL T*)uniform_u64(e.kc); p.krc = (T*)(SPATTEN_GLOBAL T*)uniform_u64(e.krc);
  p.vc = (T*)(SPATTEN_GLOBAL T*)uniform_u64(e.vc);
  p.q = (const T*)(const SPATTEN_GLOBAL T*)qp;
  p.k_new = (const T*)(const SPATTEN_GLOBAL T*)(c.append ? uniform_u64(e.kn) : qp);
  p.v_new = (const T*)(const SPATTEN_GLOBAL T*)(c.append ? uniform_u64(e.vn) : qp);
  p.out = (T*)(SPATTEN_GLOBAL T*)uniform_u64(e.out); p.scores = (T*)(SPATTEN_GLOBAL T*)uniform_u64(e.scores);
  p.ch_h = __builtin_amdgcn_readfirstlane(e.h);
  p.ws_cnt = (unsigned*)(c.ws_layers + (int64_t)l * c.ws_layer_bytes);
  p.ws_part = (unsigned long long*)(c.ws_layers + (int64_t)l * c.ws_layer_bytes + c.cnt_bytes);
  p.ch_wait = wait_l >= 0 ? c.flags + (int64_t)wait_l * c.fs : nullptr;
  p.ch_wait_n = c.base.B * wait_n;
  p.ch_done = c.flags + (int64_t)l * c.fs;
  p.ch_ny = n_active;
  p.ch_tag = tag;
  p.ch_hdr = last ? c.hdr : nullptr;
  p.ch_layer = l;
  return n_active;
}

// the rotary rows of the query's position and of the appended key's slot: the same for every layer of the token
template <typename T, int D>
__device__ __forceinline__ void chain_stage_rotary(const ChainConst<T>& c, T* s_ch) {
  using V8 = Vec8<T>;
  constexpr int HALF = D / 2, PPH = HALF / 8;
  if ((int)threadIdx.x < 4 * PPH) {
    const int which = (int)threadIdx.x / PPH, pc = (int)threadIdx.x % PPH;
    const int row = which < 2 ? min(max(c.base.pos_q, 0), c.base.table_rows - 1) : c.base.nr_row;
    const T* src = ((which & 1) ? c.base.sin : c.base.cos) + (int64_t)row * HALF + 8 * pc;
    V8::stg(s_ch + 3 * D + which * HALF + 8 * pc, V8::ldg(src));
  }
}

// `depth` lanes of workgroups per CU (1 by default), lane p serves layers p, p + depth, ...  THREADS = 512 for single-shot chunks
// (a split's whole chunk is ONE tile: Llama-2-7B decode sizes; the per-layer launch's two-waves-per-SIMD team), 256 for the
// pipelined tiles of long chunks — the per-layer launches' own instantiations of decode_body, so the bits agree.
template <typename T, int D, int UNR, bool PIPE, bool DYN, int THREADS>
__global__ __launch_bounds__(THREADS) void decode_chain_kernel(const ChainLayerDev* __restrict__ table, const ChainConst<T> c) {
  // LDS staging rows (decode_body<CHAIN>): 0 q, 1 k_new, 2 v_new of the workgroup's head in the current layer; 3 = cos | sin
  // of the query's position, 4 = cos | sin of the appended key's slot
  __shared__ __attribute__((aligned(16))) T s_ch[5 * D];
  __shared__ __attribute__((aligned(16))) ChainLds s_tab[kChainMaxLayers];
  chain_stage_rotary<T, D>(c, s_ch);
  const int lane_id = (int)blockIdx.z / c.base.B;
  const unsigned tag = c.hdr[1] + 1u;   // (the epoch only moves when every workgroup has passed this load: chain_complete)
  chain_stage_table(table, c.n_layers, s_tab);
  for (int l = lane_id; l < c.n_layers; l += c.depth) {
    SPATTEN_CSTAMP_L(l, 10);
    DecodeParams<T> p = c.base;
    if ((int)blockIdx.y >= chain_fill<T>(p, c, s_tab, l, tag)) continue;
    SPATTEN_CSTAMP_L(l, 11);
    // (Measured and dropped: a SHORT last split — a layout that gives the unit's merging split, which starts its stream last, fewer
    //  rows, run through a second instantiation with fewer row-groups per tile so that it requests nothing it does not have: 9.5 us
    //  per layer against 9.3 — the doubled loop body costs more than the shorter stream brings.)
    decode_body<T, D, UNR, 0, true, 0, true, false, PIPE, DYN, false, false, THREADS, false, 1>(p, nullptr, s_ch);
  }
}

template <typename T, int D>
static int launch_chain(const spatten_chain_args_t* a, const ChainConst<T>& c, int ny, bool pipe, hipStream_t stream) {
  constexpr int U = sizeof(T) == 4 ? 4 : 10, UP = sizeof(T) == 4 ? 2 : 4;
  const bool dyn = a->step_state != nullptr;
  const dim3 grid((unsigned)c.base.S, (unsigned)ny, (unsigned)(c.base.B * c.depth));
  const long long total = (long long)grid.x * grid.y * grid.z;
  const auto* table = (const ChainLayerDev*)a->layers;
#define SPATTEN_CHAIN_LAUNCH(KERN, TT)                                                                               \
  do {                                                                                                               \
    auto kern = KERN;                                                                                                \
    static std::atomic<int> occ{0};     /* (asked once per instantiation: not a stream operation, but not free either) */ \
    int per_cu = occ.load(std::memory_order_relaxed);                                                                \
    if (per_cu == 0) {                                                                                               \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TT, 0) != hipSuccess || per_cu <= 0) return SPATTEN_ERR_LAUNCH; \
      occ.store(per_cu, std::memory_order_relaxed);                                                                  \
    }                                                                                                                \
    /* every workgroup polls: the whole grid must be resident at once */                                             \
    if (total > (long long)per_cu * coresident_workgroups()) return SPATTEN_ERR_UNSUPPORTED;                        \
    hipLaunchKernelGGL(kern, grid, dim3(TT), 0, stream, table, c);                                                   \
  } while (0)
  if (pipe) {
    if (dyn) SPATTEN_CHAIN_LAUNCH((decode_chain_kernel<T, D, UP, true, true, kDecodeThreads>), kDecodeThreads);
    else SPATTEN_CHAIN_LAUNCH((decode_chain_kernel<T, D, UP, true, false, kDecodeThreads>), kDecodeThreads);
  } else {
    if (dyn) SPATTEN_CHAIN_LAUNCH((decode_chain_kernel<T, D, (U + 1) / 2, false, true, 2 * kDecodeThreads>), 2 * kDecodeThreads);
    else SPATTEN_CHAIN_LAUNCH((decode_chain_kernel<T, D, (U + 1) / 2, false, false, 2 * kDecodeThreads>), 2 * kDecodeThreads);
  }
#undef SPATTEN_CHAIN_LAUNCH
  return hipGetLastError() == hipSuccess ? SPATTEN_OK : SPATTEN_ERR_LAUNCH;
}

static inline size_t chain_flag_stride(int batch, int heads) { return ((size_t)batch * heads + 63) / 64 * 64; }
static inline size_t chain_layer_ws_bytes(int batch, int heads, int head_dim, int splits) {
  const size_t units = (size_t)batch * heads;
  return decode_cnt_bytes(units) + units * splits * (head_dim + 2) * sizeof(unsigned long long);
}

}  // namespace spatten

using namespace spatten;

extern "C" size_t spatten_decode_chain_workspace_bytes(int layers, int batch, int heads, int head_dim, int max_splits) {
  if (layers <= 0 || batch <= 0 || heads <= 0 || head_dim <= 0 || max_splits <= 0) return 0;
  return kDecodeWsHeader + (size_t)layers * chain_flag_stride(batch, heads) * sizeof(unsigned) +
         (size_t)layers * chain_layer_ws_bytes(batch, heads, head_dim, max_splits);
}

extern "C" int spatten_attn_decode_chain(const spatten_chain_args_t* a, void* stream) {
  if (!a || a->struct_size != sizeof(spatten_chain_args_t)) return SPATTEN_ERR_INVALID;
  if (a->n_layers > kChainMaxLayers) return SPATTEN_ERR_UNSUPPORTED;
  if (!a->layers || a->n_layers <= 0 || !a->cos || !a->sin || !a->workspace) return SPATTEN_ERR_INVALID;
  if (a->batch <= 0 || a->heads <= 0 || a->kv_len <= 0 || a->pos_q < 0 || a->workspace_splits <= 0) return SPATTEN_ERR_INVALID;
  if (a->dtype != SPATTEN_F16 && a->dtype != SPATTEN_BF16) return ok_dtype(a->dtype) ? SPATTEN_ERR_UNSUPPORTED : SPATTEN_ERR_INVALID;
  if (a->head_dim != 128) return SPATTEN_ERR_UNSUPPORTED;
  const int ny = a->max_active > 0 ? a->max_active : a->heads;
  if (ny > a->heads) return SPATTEN_ERR_INVALID;
  if (a->table_rows < a->kv_len || (!a->step_state && a->pos_q + 1 > a->table_rows)) return SPATTEN_ERR_INVALID;
  const int64_t lim = 0x7FFFFFFF;
  if (a->kv_sb > lim || a->kv_sh > lim) return SPATTEN_ERR_UNSUPPORTED;
  const int depth = a->depth > 0 ? a->depth : 1;
  if (depth > 4) return SPATTEN_ERR_INVALID;
  // the split-N decomposition: decode_rows' own (decode_attn.hip), so a chained token adds its partials in the order of the
  // per-layer launches of the same shape — bit-identical outputs
  const int lay = (!a->step_state && a->kv_len_layout > a->kv_len) ? a->kv_len_layout : a->kv_len;
  int S = a->n_splits > 0 ? a->n_splits : spatten_decode_auto_splits(a->batch, ny, a->head_dim, lay);
  if (S > lay) S = lay;
  if (S > kDecodeMaxSplits) S = kDecodeMaxSplits;
  const int chunk = ceil_div(ceil_div(lay, S), 8) * 8;
  S = ceil_div(lay, chunk);
  if (S > a->workspace_splits) return SPATTEN_ERR_INVALID;
  const bool pipe = chunk > 10 * (kDecodeThreads / (a->head_dim / 16));
  if (!pipe && decode_team() != 512) return SPATTEN_ERR_UNSUPPORTED;   // (the 256-thread single-shot team is not instantiated here)
  const int d = a->head_dim;
  const size_t fs = chain_flag_stride(a->batch, a->heads);
  char* base = (char*)a->workspace;
#define SPATTEN_CHAIN_FILL(T)                                                                                   \
  {                                                                                                             \
    ChainConst<T> c;                                                                                            \
    memset(&c, 0, sizeof(c));                                                                                   \
    DecodeParams<T>& p = c.base;                                                                                \
    p.q_sb = (int64_t)a->heads * d; p.q_sh = d; p.kv_sb = a->kv_sb; p.kv_sh = a->kv_sh;                         \
    p.new_sb = a->append ? a->new_sb : p.q_sb; p.new_sh = a->append ? a->new_sh : p.q_sh; p.append = a->append ? 1 : 0; \
    p.cos = (const T*)a->cos; p.sin = (const T*)a->sin; p.table_rows = a->table_rows;                          \
    p.step = (const int32_t*)a->step_state; p.nr_row = (a->kv_len < a->table_rows ? a->kv_len : a->table_rows) - 1; \
    if (a->step_state) {                                                                                        \
      p.cos = (const T*)((const char*)a->step_state + kStepHeader);                                            \
      p.sin = p.cos + 2 * (d / 2); p.table_rows = 2; p.nr_row = 1;                                              \
    }                                                                                                           \
    p.out_sb = a->out_sb; p.sc_sb = a->sc_sb; p.sc_sh = a->sc_sh; p.lse_q = 1;                                  \
    p.ws_err = (unsigned*)base; p.ws_unit = (int64_t)a->workspace_splits * (d + 2);                             \
    p.B = a->batch; p.H = a->heads; p.Hkv = a->heads; p.N = a->kv_len; p.pos_q = a->step_state ? 0 : a->pos_q;  \
    p.S = S; p.chunk = chunk; p.n_q = 1; p.vis0 = a->kv_len; p.poll_merge = S > 1 ? 1 : 0;                      \
    p.sqrt_d = sqrtf((float)d);                                                                                 \
    c.n_layers = a->n_layers; c.depth = depth; c.append = a->append ? 1 : 0;                                    \
    c.hdr = (unsigned*)base; c.flags = (unsigned*)(base + kDecodeWsHeader); c.fs = (int)fs;                     \
    c.ws_layers = base + kDecodeWsHeader + (size_t)a->n_layers * fs * sizeof(unsigned);                         \
    c.ws_layer_bytes = (int64_t)chain_layer_ws_bytes(a->batch, a->heads, d, a->workspace_splits);               \
    c.cnt_bytes = (int64_t)decode_cnt_bytes((size_t)a->batch * a->heads);                                       \
    return launch_chain<T, 128>(a, c, ny, pipe, (hipStream_t)stream);                                           \
  }
  if (a->dtype == SPATTEN_F16) SPATTEN_CHAIN_FILL(f16_t)
  SPATTEN_CHAIN_FILL(bf16_t)
#undef SPATTEN_CHAIN_FILL
}

#ifdef SPATTEN_CHAIN_TRACE
extern "C" int spatten_debug_set_chain_trace(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
