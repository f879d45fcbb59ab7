/*
 * spatten.h — C ABI of libspatten_hip.so: the MI355X (gfx950) implementation of the
 * SpAtten cascade-pruned attention hot path.
 *
 * Every entry point takes plain device pointers, explicit shapes/strides (in ELEMENTS), a dtype
 * enum and a hipStream_t (passed as void*).  Functions never allocate, never synchronise and never
 * throw; they return 0 on success or a negative spatten_status_t.  All work is enqueued on `stream`.
 *
 * Reference interfaces each entry point replaces (paths relative to mit-han-lab/spatten):
 *
 *   spatten_attn_decode      spatten_llm/pos_shift/modify_llama.py:86-147 at q_len == 1
 *                            (cache-relative RoPE of Q and of the whole un-rotated K cache, KV append,
 *                            Q.K^T/sqrt(d), score stash :116-119, +mask, fp32 softmax, P.V, head merge)
 *   spatten_attn_prefill     the same lines at q_len > 1 (causal or explicit additive mask)
 *   spatten_rope_single      spatten_llm/pos_shift/modify_llama.py:21-28 (apply_rotary_pos_emb_single)
 *   spatten_importance       spatten_llm/kv_cache_token_pruning.py:51   (stash.sum(0).sum(1))
 *   spatten_topk_select      spatten_llm/kv_cache_token_pruning.py:59-63 (window top-k, sort, +start)
 *   spatten_kv_compact       spatten_llm/kv_cache_token_pruning.py:64-96 (mask gather + 3-way concat)
 *   spatten_prune_layers     the per-layer loop kv_cache_token_pruning.py:55-96 as ONE batched launch pair
 *
 * Entry points with no numeric counterpart in the reference ("parity unpinned": restated from the
 * RTL control flow): spatten_importance_accumulate (cascade importance),
 * spatten_head_scores, spatten_pq_pack / the pq_* fields of spatten_attn_decode_args (progressive quantisation).
 */
#ifndef SPATTEN_H_
#define SPATTEN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPATTEN_ABI_VERSION 5

typedef enum {
  SPATTEN_F32 = 0,
  SPATTEN_F16 = 1,
  SPATTEN_BF16 = 2
} spatten_dtype_t;

typedef enum {
  SPATTEN_OK = 0,
  SPATTEN_ERR_INVALID = -1,     /* bad pointer / shape / stride / dtype */
  SPATTEN_ERR_UNSUPPORTED = -2, /* head_dim or size outside the compiled range */
  SPATTEN_ERR_WINDOW = -3,      /* top-k window holds fewer than k candidates (reference: torch.topk RuntimeError) */
  SPATTEN_ERR_LAUNCH = -4,      /* hipLaunch failed; see hipGetLastError on the caller side */
  SPATTEN_ERR_TIMEOUT = -5      /* a kernel gave up waiting for another workgroup's data (device error word set) */
} spatten_status_t;

int spatten_abi_version(void);
const char* spatten_status_string(int status);

/* ------------------------------------------------------------------------------------------------
 * Workspace.  Split-N decode keeps per-(b,h,split) partials (fp32) and one arrival counter per
 * (b,h).  The caller allocates `spatten_decode_workspace_bytes` bytes ONCE, zero-fills it once
 * (hipMemset) and may reuse it for every launch on the same stream (the kernel re-arms the
 * counters itself).  Layout: a 256-byte header (word 0: device error flag), the counters, then one
 * FIXED region of max_splits partials per (b,h) — launches with different split counts, head
 * subsets or key sources never alias another unit's region.
 * ---------------------------------------------------------------------------------------------- */
#define SPATTEN_DECODE_MAX_SPLITS 64
size_t spatten_decode_workspace_bytes(int batch, int heads, int head_dim, int max_splits);
/* Split count the library would pick for this shape (>=1).  `n_splits` <= 0 in the calls below means "auto". */
int spatten_decode_auto_splits(int batch, int heads, int head_dim, int kv_len);
/* The ONE synchronising call of the library: waits for `stream`, reads the workspace's error flag back and clears
 * it.  SPATTEN_OK, or SPATTEN_ERR_TIMEOUT when a split-N merge expired its (bounded) wait for a partial — the outputs
 * of that launch were poisoned with NaN instead of being merged from incomplete data. */
int spatten_decode_workspace_status(void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Decode attention (q_len == 1), fused:  modify_llama.py:86-147.
 *
 *   q        [B, H, d]            un-rotated query of the new token      (strides q_sb, q_sh; d contiguous)
 *   k_cache  [B, Hkv, cap, d]     UN-rotated keys (modify_llama.py:100)  (strides kv_sb, kv_sh; rows contiguous,
 *                                 pitch d).  Only WRITTEN (append of the new row); may be NULL when k_new is NULL
 *   kr_cache [B, Hkv, cap, d]     rotated shadow: row j = apply_rotary_pos_emb_single(k_cache row j, position j)
 *                                 (modify_llama.py:103-104), valid for rows [0, kv_len - (k_new ? 1 : 0)); built with
 *                                 spatten_rope_single, kept current by this call's append.  Same strides
 *   v_cache  [B, Hkv, cap, d]     values, same strides
 *   k_new, v_new [B, Hkv, d]      optional (may be NULL): the new token's un-rotated K/V rows; when given they
 *                                 are appended IN PLACE at slot kv_len-1 of k_cache / v_cache (replaces torch.cat,
 *                                 :95-98), the rotated row goes to kr_cache, and they are used for that slot;
 *                                 strides new_sb, new_sh
 *   cos, sin [table_rows >= max(kv_len, pos_q+1), d/2]  rotary table in the model dtype (transformers 4.33
 *                                 LlamaRotaryEmbedding rounds it with .to(x.dtype)); only the first half of
 *                                 the d columns is stored because emb = cat(freqs, freqs)
 *   table_rows  number of rows of cos/sin (positions are clamped to it for memory safety)
 *   pos_q    rotary position of the query (HF passes past_len); key j is rotated at position j (:103-104)
 *   position_ids optional int64 [B] (stride pos_sb) DEVICE pointer: per-batch query position, overrides
 *            pos_q (what HF hands the forward as position_ids[:, 0]; no host sync needed to honour it)
 *   mask     optional additive mask [B, kv_len] in the model dtype (the [B,1,1,N] HF mask squeezed), stride mask_sb
 *   out      [B, H*d]             attn_output before o_proj, model dtype (:138-147), stride out_sb
 *   scores   optional [B, H, kv_len] the stash: raw scaled logits BEFORE mask/softmax, rounded like the
 *                                 reference (matmul -> dtype, /sqrt(d) -> dtype) (:111-119), strides sc_sb, sc_sh
 *   lse      optional [B, H, 2] fp32: (row max of masked logits, sum exp(logit - max))  -> max prob = 1/sum
 *   workspace  sized with spatten_decode_workspace_bytes(batch, heads, head_dim, SPATTEN_DECODE_MAX_SPLITS)
 * ---------------------------------------------------------------------------------------------- */
int spatten_attn_decode(int dtype,
                        const void* q, int64_t q_sb, int64_t q_sh,
                        void* k_cache, void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh,
                        const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh,
                        const void* cos, const void* sin, int table_rows,
                        const int64_t* position_ids, int64_t pos_sb,
                        const void* mask, int64_t mask_sb,
                        void* out, int64_t out_sb,
                        void* scores, int64_t sc_sb, int64_t sc_sh,
                        float* lse,
                        void* workspace,
                        int batch, int heads, int kv_heads, int head_dim,
                        int kv_len, int pos_q, int n_splits,
                        void* stream);

/* The same decode step with every SpAtten extra, as ONE argument block (zero-fill it, set struct_size, fill what
 * you use; fields up to and including n_splits mean exactly what the positional form documents):
 *   workspace_splits           the max_splits the workspace was sized with (0 = SPATTEN_DECODE_MAX_SPLITS)
 *   head_ids / n_active_heads  head pruning: only the listed query heads are launched (ascending int32 list in
 *                              DEVICE memory, NULL = all); rows of pruned heads in out / scores / lse are left untouched
 *   flags                      SPATTEN_DECODE_SCORES_ONLY: write the stash and lse only (no V traffic, no output) —
 *                              pass 1 of local V pruning (scores, lse required; k_new must be NULL)
 *   importance_acc ...         cascade (cumulative) importance, README.md:11, fused and deferred by one step: while this
 *                              step's keys stream, acc[h, j] += exp(prev_scores[b,h,j] - prev_lse[b,h,0]) / prev_lse[b,h,1]
 *                              for j < prev_len — the softmax probabilities of the PREVIOUS decode step, whose stash and
 *                              (max, sum) were written by that step's launch (strides prev_sb, prev_sh; acc [H, >=prev_len]
 *                              fp32, stride acc_sh).  The last step before a prune is folded by spatten_importance_accumulate.
 *                              Needs mask == NULL (a single-token step sees every key).
 *   head_abs_acc               head pruning (README.md:21): fp32 [B*H]; head_abs_acc[b*H+h] += sum_e |out[b, h*d+e]| of the value
 *                              this call leaves in `out` (deterministic order) — the cumulative head importance
 *   pq_*                       progressive quantisation (MatrixFetcher.scala:48-51,341-348; RequantDecision.scala:44-72;
 *                              SpAttenController.scala:35-39,402): keys come from the MSB / LSB planes written by
 *                              spatten_pq_pack instead of kr_cache (two launches of the decode kernel: pass 1 = logits from
 *                              the MSB plane, softmax, P.V with the un-quantised V, need_lsb = max_j prob_j < threshold per
 *                              head; pass 2 = flagged heads refetch the LSB plane and are recomputed once, confident heads
 *                              return at once).  pq_need_lsb: int32 [B*H] scratch / output (required).  k_new and flags
 *                              must be NULL / 0 in this mode (append with spatten_kv_append + spatten_pq_pack first);
 *                              `scores` then holds the logits of the quantised keys that the output was computed from. */
#define SPATTEN_DECODE_SCORES_ONLY 1
typedef struct spatten_decode_args {
  uint32_t struct_size;            /* sizeof(spatten_decode_args_t) of the caller */
  int32_t dtype;
  const void* q; int64_t q_sb, q_sh;
  void* k_cache; void* kr_cache; void* v_cache; int64_t kv_sb, kv_sh;
  const void* k_new; const void* v_new; int64_t new_sb, new_sh;
  const void* cos; const void* sin; int32_t table_rows; int32_t pad0_;
  const int64_t* position_ids; int64_t pos_sb;
  const void* mask; int64_t mask_sb;
  void* out; int64_t out_sb;
  void* scores; int64_t sc_sb, sc_sh;
  float* lse;
  void* workspace; int32_t workspace_splits;
  int32_t batch, heads, kv_heads, head_dim, kv_len, pos_q, n_splits;
  const int32_t* head_ids; int32_t n_active_heads; int32_t flags;
  const void* prev_scores; int64_t prev_sb, prev_sh; const float* prev_lse; int32_t prev_len; int32_t pad1_;
  float* importance_acc; int64_t acc_sh;
  float* head_abs_acc;
  const void* pq_msb; const void* pq_lsb; const float* pq_scale;
  int64_t pq_pl_sb, pq_pl_sh, pq_sc_sb, pq_sc_sh;
  float pq_threshold; int32_t pad2_;
  int32_t* pq_need_lsb;
  const void* step_state;          /* ABI 3: device-resident cache length (see "Device-resident step state" below) */
  int32_t kv_len_layout;           /* ABI 3: 0 = kv_len.  Otherwise >= kv_len: the split-N decomposition (splits, rows per
                                      split) is laid out for THIS length, so steps at different lengths share it — a static
                                      launch and a step_state launch whose bound (its kv_len) is this value add their
                                      partials in the same order: bit-identical outputs.  Later splits may then be empty. */
  int32_t pad3_;
  /* ABI 3: the output projection of the step (modify_llama.py:163) in the same CALL: proj_out[b, n] = sum_k out[b, k] *
     proj_weight[n, k] (+ proj_bias[n]); weight [proj_n, heads*head_dim] row stride proj_w_sn (elements), output row stride
     proj_out_sb.  `out` is still written; the projection is a second launch (spatten_gemv's kernel: bit-identical with
     spatten_gemv(out)) issued by the same call — one host call per layer-step instead of two. */
  const void* proj_weight; int64_t proj_w_sn; const void* proj_bias; void* proj_out; int64_t proj_out_sb; int32_t proj_n;
  int32_t pad4_;
  /* ABI 4: the step's q / k / v projections (modify_llama.py:72-74) INSIDE the attention launch: qkv_x [hidden] = the
     layer's input row, qkv_weight [3*heads*head_dim, hidden] = q_proj / k_proj / v_proj stacked (row stride qkv_w_sn),
     qkv_bias optional [3*heads*head_dim], qkv_exchange = spatten_decode_qkv_exchange_bytes() of device scratch (zero-filled
     once, one per workspace).  q, k_new, v_new must then be NULL: each workgroup projects its share of the head's q / k / v
     while the K/V stream of the step is already in flight (the values equal spatten_gemv's bit for bit).  Only where
     spatten_decode_qkv_supported() says so; SPATTEN_ERR_UNSUPPORTED otherwise (launch spatten_gemv + the plain step).
     With proj_* set as well, the output projection runs inside the SAME launch when proj_n = 16 x the launch's workgroups
     and heads*head_dim = 4096 (Llama-2-7B: the whole attention module of a decode step is one launch; same bits as
     spatten_gemv(out)), as the second launch of the call otherwise. */
  const void* qkv_x; const void* qkv_weight; int64_t qkv_w_sn; const void* qkv_bias; void* qkv_exchange; int32_t qkv_hidden;
  int32_t pad5_;
} spatten_decode_args_t;
int spatten_attn_decode_args(const spatten_decode_args_t* args, void* stream);
/* the fused q/k/v projection + attention launch (spatten_decode_args_t::qkv_*): scratch size, and whether a step of this
 * shape runs it (lean MHA step, bf16 / f16, head_dim 128, batch 1, <= 320 rows per split of the layout length) */
size_t spatten_decode_qkv_exchange_bytes(int batch, int heads, int head_dim);
/* Threads of the attention team of a single-row, single-shot decode step (ABI 4, round 4): 512 = two waves per SIMD (the
 * default: 11.5 -> 11.0 us at the Llama-2-7B headline shape), 256 = the r03 form.  Process-wide; every route of such a step
 * (lean / general, cascade accumulation, device length) follows it, so results stay bit-identical between routes; the two
 * forms differ from each other in summation order (low bits).  The fused projection launch (qkv_*) contains the 256-thread
 * body: a caller that wants its fused and separate steps bit-identical selects 256.  Returns the previous value (or
 * SPATTEN_ERR_INVALID).  SPATTEN_DECODE_TEAM=256 in the environment sets the initial value. */
int spatten_decode_set_team(int threads);
int spatten_decode_qkv_supported(int dtype, int batch, int heads, int kv_heads, int head_dim, int kv_len_layout);
/* Grouped-query models (heads > kv_heads; modify_llama.py:106-108 repeat_kv), round 6: the single-row step of bf16 / f16, head_dim
 * 128, without mask / head list / cascade accumulation / quantised keys streams a KV head's rows ONCE for its whole query group and
 * scores them on the matrix cores (csrc/decode_gqa.hip) instead of once per query head.  mode: -1 = where that form measured faster
 * (the default: a cost model fitted to 27 measured shapes, csrc/decode_gqa.hip gqa_pays — 32 / 8 heads from ~3k rows, 64 / 8 from
 * ~1.5k; the layout length decides, so a static launch and the device-length form of one step take the same kernel), 0 = never, 1 = whenever the
 * step is eligible.  Process-wide; outputs of the two forms differ in summation order (low bits), both
 * inside the stated tolerance.  Returns the previous mode + 1 (0..2), or SPATTEN_ERR_INVALID.  SPATTEN_DECODE_GQA in the
 * environment sets the initial mode. */
int spatten_decode_set_gqa(int mode);
/* 1 when a plain single-row step of this geometry (dtype bf16 / f16, head_dim 128, heads > kv_heads, kv_len_layout = the layout length:
 * the cache length of a static launch, the bound of the device-length form) is launched in the matrix-core form under the current
 * mode, 0 otherwise (also for geometries the form does not serve).  No stream operation; the default mode's answer follows the
 * device's CU count. */
int spatten_decode_gqa_selected(int dtype, int batch, int heads, int kv_heads, int head_dim, int kv_len_layout);

/* ------------------------------------------------------------------------------------------------
 * The chained decode launch (ABI 5, round 6): the attention step of ALL layers of one token — what the caller's per-layer
 * loop issues as n_layers spatten_attn_decode_args launches (modify_llama.py:86-147 once per LlamaAttention module) — in ONE
 * launch.  At batch 1 a layer's launch is latency-bound; a layer's K/V rows depend on nothing upstream, only its query and
 * appended row do (q_l is a function of out_{l-1}, modify_llama.py:72-92).  One workgroup per (split, head) WALKS the layers:
 * at the top of a layer's step its waves request their rows of the K/V tile at once; wave 0 alone first waits for the
 * completion words of the previous layer (one per (batch, head), stored by that unit's merger behind its `out` row), fetches
 * q / k_new / v_new into LDS for the others, and only then requests its own rows (a wave's loads return in order: a poll behind
 * the tile would see the flag a whole stream late).  The kernel boundary, the first-byte latency and the ramp of layer l + 1's
 * stream thereby run under layer l's reduce / publish / merge tail, while the layer dependency of the real model is kept.
 * Results (`out`, stash, appended rows) are bit-identical to the per-layer launches of the same shape.
 *   layers          DEVICE table [n_layers] of spatten_chain_layer_t (read by the kernel: a captured graph needs no argument
 *                   patching; the caller rewrites entries between tokens with ordinary stream operations if pointers move)
 *   per layer       k_cache (optional) / kr_cache / v_cache [B,H,cap,d] (strides kv_sb, kv_sh: shared by all layers),
 *                   q [B,H,d] DENSE, k_new / v_new [B,H,d] (strides new_sb, new_sh; read when `append` != 0), out [B,H*d]
 *                   (stride out_sb), scores optional stash [B,H,>=kv_len] (strides sc_sb, sc_sh), head_ids optional int32
 *                   [n_active] (head pruning / a head-parallel rank's survivors; NULL = heads 0..n_active-1), n_active heads
 *                   launched in this layer (0: the layer is skipped, the dependency passes through it)
 *   kv_len, pos_q, kv_len_layout, step_state, n_splits   as in spatten_decode_args_t — the same for every layer of the token
 *   max_active      grid columns: max over the layers of n_active (0 = heads)
 *   depth           lanes of workgroups, 1..4 (0 = 1): lane p serves layers p, p + depth, ... (more than one lane only for
 *                   the pipelined tiles of long chunks — the single-shot form needs the whole register file of a CU)
 *   flags           reserved (0).  A completion word is ordered after its `out` row only for observers that wait for the
 *                   kernel boundary — the launch's own layers need nothing more: they read q, not out (the word stands where a
 *                   fused producer of q would publish its result).
 *   workspace       spatten_decode_chain_workspace_bytes(n_layers, batch, heads, head_dim, workspace_splits) bytes,
 *                   zero-filled once, one per stream; spatten_decode_workspace_status() reports a timed-out wait (the
 *                   launch then ran to its end on incomplete data: zero-fill the workspace again before the next token — the
 *                   token epoch did not advance)
 * MHA, bf16 / f16, head_dim 128, the lean step (no mask / position tensor / cascade accumulation); every workgroup polls, so the
 * whole grid (splits x max_active x batch x depth) must be co-resident: SPATTEN_ERR_UNSUPPORTED otherwise (launch the layers
 * one by one), also under spatten_decode_set_team(256).
 * ---------------------------------------------------------------------------------------------- */
typedef struct spatten_chain_layer {
  void* k_cache; void* kr_cache; void* v_cache;
  const void* q; const void* k_new; const void* v_new;
  void* out; void* scores;
  const int32_t* head_ids; int32_t n_active; int32_t pad_;
} spatten_chain_layer_t;                 /* 80 bytes */
typedef struct spatten_chain_args {
  uint32_t struct_size;                  /* sizeof(spatten_chain_args_t) of the caller */
  int32_t dtype;
  const void* layers;                    /* DEVICE pointer: spatten_chain_layer_t [n_layers] */
  int32_t n_layers, depth;
  int64_t kv_sb, kv_sh, new_sb, new_sh, out_sb, sc_sb, sc_sh;
  const void* cos; const void* sin; int32_t table_rows; int32_t append;
  void* workspace; int32_t workspace_splits;
  int32_t batch, heads, head_dim, kv_len, pos_q, n_splits, max_active, flags, kv_len_layout;
  const void* step_state;
} spatten_chain_args_t;
size_t spatten_decode_chain_workspace_bytes(int layers, int batch, int heads, int head_dim, int max_splits);
int spatten_attn_decode_chain(const spatten_chain_args_t* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-resident step state (ABI 3) — what makes a decode step capturable ONCE and replayable for every token of a
 * turn.  The reference's caller is a per-token Python loop (run_spatten_llama.py:27-35) in which the cache length is a
 * host integer that changes every token; with `step_state` set in spatten_decode_args_t the kernel reads it from
 * device memory instead:
 *   - args.kv_len becomes a BOUND: the grid, the split chunks and every load address are laid out for it; it must be
 *     >= the device value at every replay and <= the capacity of the planes and of the stash row.  args.pos_q is ignored.
 *   - the length AFTER this step's append is state word 0; the query's rotary position is state word 1; the rotary rows
 *     of those two positions are staged in the state (copied from the tables by spatten_step_set / _advance), so no
 *     load address in the attention kernel depends on the length.
 *   - rows [length, bound) of kr_cache and v_cache are read and discarded (weight 0): they must hold FINITE values —
 *     zero-fill the planes once when they are allocated.  Stash entries [length, bound) are left untouched.
 *   - admitted for the single-token step without mask, position_ids and SCORES_ONLY; head_ids / head_abs_acc as usual.
 *   - pq_* (progressive-quant keys) are admitted (with or without importance_acc): the step's row is appended beforehand by
 *     spatten_kv_append_step (which also packs its planes); plane rows [length, bound) must be finite too.
 *   - importance_acc (the fused cascade accumulation) IS admitted: `scores` / `lse` and `prev_scores` / `prev_lse` are
 *     then the TWO buffers that swap roles every step (same strides, rows up to the bound): step k since the last
 *     spatten_step_set writes buffer (k - 1) & 1 — `scores` first — and folds the other one over the rows the previous
 *     step wrote (state word 3; 0 for the first step).  prev_len is ignored.
 * A token of a captured graph = spatten_step_advance(state, ..., 1) followed by the layers' spatten_attn_decode_args
 * launches sharing the state; before the first replay of a turn: spatten_step_set(state, ..., cache_len, cache_len - 1).
 * The state is spatten_step_state_bytes(dtype, head_dim) bytes of device memory owned by the caller.
 * ---------------------------------------------------------------------------------------------- */
size_t spatten_step_state_bytes(int dtype, int head_dim);
/* state := (kv_len, pos_q) from HOST values + the rotary rows of positions pos_q and kv_len - 1 (cos / sin: the half tables
 * [table_rows, d/2]; positions are clamped to the table).  Negative kv_len = "no step yet" is not accepted: kv_len >= 0. */
int spatten_step_set(void* state, int dtype, int head_dim, const void* cos, const void* sin, int table_rows,
                     int kv_len, int pos_q, void* stream);
/* state := (kv_len + delta, pos_q + delta) + the rotary rows of the new positions — a stream operation without host
 * values of the length, hence capturable. */
int spatten_step_advance(void* state, int dtype, int head_dim, const void* cos, const void* sin, int table_rows,
                         int delta, void* stream);

/* The append of ONE decode step in device-length form, for the modes whose attention launch appends nothing (pq_*):
 * row (state word 0) - 1 of k_cache (optional) / v_cache <- k_new / v_new [B,Hkv,d] (strides new_sb, new_sh), of kr_cache
 * <- k_new rotated with the state's staged row of that slot (modify_llama.py:95-104), and — msb / lsb / scale given, or
 * all three NULL — that row of the progressive-quant planes (what spatten_kv_append + spatten_pq_pack leave there, bit for
 * bit).  Call it after the token's spatten_step_advance; `capacity` = rows of the cache planes AND of the quantised planes:
 * a row >= capacity is not written.  head_dim 64 / 128. */
int spatten_kv_append_step(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh, void* k_cache,
                           void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, void* msb, void* lsb, float* scale,
                           int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh, int batch, int kv_heads,
                           int head_dim, int capacity, const void* step_state, void* stream);

/* The projections of a single-token step (modify_llama.py:72-74 q/k/v_proj, :163 o_proj; nn.Linear semantics):
 *   y[m, n] = sum_k x[m, k] * W[n, k] (+ bias[n]),  W [N, K] row-major with row stride w_sn (elements), x [M, K] row stride
 *   x_sm, y [M, N] row stride y_sm, bias optional [N]; fp32 accumulation, one rounding to the dtype.  A weight-streaming
 *   kernel for M = 1 (a batch of M single-token rows streams W once per row); K, w_sn, x_sm multiples of 8.
 *   Multi-token forwards keep their GEMM library. */
int spatten_gemv(int dtype, const void* x, int64_t x_sm, const void* W, int64_t w_sn, const void* bias, void* y,
                 int64_t y_sm, int M, int N, int K, void* stream);

/* KV append without attention (modify_llama.py:95-100 + the shadow row): k_new / v_new [B,Hkv,n,d] (strides new_sb,
 * new_sh, new_sn; d contiguous) are written to rows [row0, row0+n) of k_cache / v_cache, and their rotation at slot
 * positions row0+i to kr_cache (all three [B,Hkv,cap,d], strides kv_sb, kv_sh, rows contiguous).  Used by the modes
 * whose attention launch does not append (progressive quantisation, local V pruning) and by prefill. */
int spatten_kv_append(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh, int64_t new_sn,
                      void* k_cache, void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh,
                      const void* cos, const void* sin, int table_rows,
                      int batch, int kv_heads, int n, int head_dim, int row0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Prefill attention (q_len >= 1), flash-style, same semantics as above for a block of queries.
 * bf16/f16 with head_dim 64/128: rotate Q into the workspace, lay V out key-contiguous for the matrix cores,
 * then MFMA flash attention with online softmax.  fp32, q_len <= 8 or head_dim 256: the decode kernel, one
 * softmax row per workgroup column (exact fp32).
 *
 *   q            [B, H, q_len, d]   un-rotated (strides q_sb, q_sh, q_sq; d contiguous) — the [B,q,H*d]
 *                                   projection output viewed as [B,H,q,d] is accepted without a copy
 *   kr_cache     [B,Hkv,cap,d]      rotated shadow of the key cache (see spatten_attn_decode), rows [0, kv_len)
 *   v_cache      [B,Hkv,cap,d]      values; both must ALREADY hold the q_len new rows at [kv_len-q_len, kv_len)
 *   pos_q0       rotary position of query row 0 (row i uses pos_q0 + i) unless
 *   position_ids optional int64 [B, q_len] DEVICE pointer (stride pos_sb, rows contiguous)
 *   causal       flag word.  Bit 0 (1): HF causal mask (key j visible to row i iff j <= kv_len - q_len + i); 0: none.
 *                Bit 1 (SPATTEN_PREFILL_FAST_NUMERICS): opt OUT of the reference's two 16-bit roundings of every logit
 *                (matmul -> dtype, / sqrt(d) -> dtype, modify_llama.py:111-113): logits stay fp32 and the scale is folded
 *                into the exponent — a faster softmax phase whose output stays within the stated tolerance of the
 *                reference EXCEPT where logits are large (one 16-bit ulp of a logit of magnitude 20-30 is a 13-28 % change
 *                of its probability).  Honoured by the MFMA leg without mask / scores / col_importance / lse (those are
 *                defined on the rounded logits); ignored elsewhere.
 *   mask         optional additive [B, q_len, kv_len] in the model dtype (the [B,1,q,N] HF mask), strides
 *                mask_sb, mask_sq; applied in addition to `causal`
 *   out          [B, q_len, H*d]    (strides out_sb, out_sq)
 *   scores       optional [B,H,q_len,kv_len] stash (strides sc_sb, sc_sh, sc_sq) — pre-mask, like the reference
 *   col_importance optional [B,H,kv_len] fp32, ZERO-FILLED by the caller: += sum over query rows of the
 *                stash column = the reference importance (kv_cache_token_pruning.py:51) without the stash
 *                (MFMA leg only; fp32 atomics, so the summation order is not reproducible run to run)
 *   lse          optional [B,H,q_len,2] fp32 (contiguous): per query row (reference max m, sum_j exp(logit_ij - m)) of the
 *                masked logits — the softmax statistics; input of spatten_importance_accumulate_prefill
 *   workspace    spatten_prefill_workspace_bytes(...) bytes of device scratch (the key-contiguous copy of V; for short
 *                query blocks on a long cache — few query blocks x heads — also the fp32 partials of the KEY SPLIT:
 *                up to 8 workgroups per (b, h, 256-query block), each over a range of key tiles, folded by a merge launch)
 * ---------------------------------------------------------------------------------------------- */
#define SPATTEN_PREFILL_FAST_NUMERICS 2
size_t spatten_prefill_workspace_bytes(int dtype, int batch, int heads, int kv_heads, int head_dim,
                                       int q_len, int kv_len);
int spatten_attn_prefill(int dtype,
                         const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq,
                         const void* kr_cache, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                         const void* cos, const void* sin, int table_rows,
                         const int64_t* position_ids, int64_t pos_sb,
                         const void* mask, int64_t mask_sb, int64_t mask_sq,
                         void* out, int64_t out_sb, int64_t out_sq,
                         void* scores, int64_t sc_sb, int64_t sc_sh, int64_t sc_sq,
                         float* col_importance, float* lse,
                         void* workspace,
                         int batch, int heads, int kv_heads, int head_dim,
                         int q_len, int kv_len, int pos_q0, int causal,
                         void* stream);

/* apply_rotary_pos_emb_single (modify_llama.py:21-28): x [B,H,n,d] (strides x_sb, x_sh, x_sn; d contiguous)
 * -> y [B,H,n,d] (strides y_sb, y_sh, y_sn); position_ids int64 [B,n] DEVICE pointer (stride pos_sb; pass
 * pos_sb = 0 to broadcast one row; NULL = positions pos0 + i).  cos/sin are the [table_rows, d/2] half tables. */
int spatten_rope_single(int dtype, const void* x, int64_t x_sb, int64_t x_sh, int64_t x_sn,
                        void* y, int64_t y_sb, int64_t y_sh, int64_t y_sn,
                        const void* cos, const void* sin, int table_rows,
                        const int64_t* position_ids, int64_t pos_sb, int pos0,
                        int batch, int heads, int n, int head_dim, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Prune event.
 * ---------------------------------------------------------------------------------------------- */
/* importance = stash.sum(0).sum(1): stash [B,H,q,L] (strides sb, sh, sq; L contiguous) -> out [H,L] (stride out_sh),
 * model dtype, each of the two reductions accumulated in fp32 and rounded to the dtype (kv_cache_token_pruning.py:51). */
int spatten_importance(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq,
                       void* out, int64_t out_sh, int batch, int heads, int q_len, int kv_len, void* stream);

/* Per head h: the k largest of score[h, lo:hi) (ties at the k-th value: lowest index first), returned as
 * ASCENDING absolute positions idx[h, 0..k) (int32, stride idx_sh).  score [H, >=hi] (stride score_sh).
 * NaN ranks largest, -0 == +0 (torch.topk order).  Requires hi - lo >= k > 0 (else SPATTEN_ERR_WINDOW). */
int spatten_topk_select(int dtype, const void* score, int64_t score_sh, int heads,
                        int lo, int hi, int k, int32_t* idx, int64_t idx_sh, void* stream);

/* Fused gather + concat (kv_cache_token_pruning.py:64-96): for X in (K, V)
 *   dst[b,h,0:start]              = src[b,h,0:start]
 *   dst[b,h,start:start+k]        = src[b,h,idx[h,:]]
 *   dst[b,h,start+k:start+k+tail] = src[b,h,tail_lo:tail_lo+tail_len]
 * src [B,H,L,d] (strides src_sb, src_sh), dst [B,H,cap',d] (strides dst_sb, dst_sh), rows contiguous (pitch d).
 * v_src/v_dst may be NULL to move K only.  kr_dst (optional, same strides as k_dst) receives the rotated shadow
 * of the NEW cache, row r rotated at position r (modify_llama.py:103-104) with the half tables cos/sin
 * [table_rows >= new length, d/2] — the slots moved, so the shadow is rebuilt by the pass that moves them. */
int spatten_kv_compact(int dtype, const void* k_src, const void* v_src, int64_t src_sb, int64_t src_sh,
                       void* k_dst, void* v_dst, void* kr_dst, int64_t dst_sb, int64_t dst_sh,
                       const void* cos, const void* sin, int table_rows,
                       const int32_t* idx, int64_t idx_sh,
                       int batch, int heads, int head_dim,
                       int start, int k, int tail_lo, int tail_len, void* stream);

/* The whole per-layer loop of apply_token_pruning (kv_cache_token_pruning.py:55-96) for `layers` layers in
 * two launches (select, then gather).  Arrays of `layers` device pointers, themselves in DEVICE memory:
 *   score_ptrs[l] -> [H, >=hi] model dtype (importance of layer l, stride score_sh)
 *   k_src_ptrs/v_src_ptrs[l] -> [B,H,L,d] ; k_dst_ptrs/v_dst_ptrs[l] -> [B,H,cap',d]
 *   idx [layers, H, k] int32 scratch/output (contiguous). */
int spatten_prune_layers(int dtype, int layers,
                         const void* const* score_ptrs, int64_t score_sh,
                         const void* const* k_src_ptrs, const void* const* v_src_ptrs, int64_t src_sb, int64_t src_sh,
                         void* const* k_dst_ptrs, void* const* v_dst_ptrs, void* const* kr_dst_ptrs /* optional */,
                         int64_t dst_sb, int64_t dst_sh,
                         const void* cos, const void* sin, int table_rows /* for kr_dst_ptrs */,
                         int32_t* idx,
                         int batch, int heads, int head_dim,
                         int lo, int hi, int k, int tail_lo, int tail_len, void* stream);

/* The same event ranked by scores of their OWN dtype (cascade importance: fp32 accumulators, README.md:11) while the
 * cache keeps the model dtype, with the accumulators carried through the row map by a third all-layer launch:
 *   score_ptrs[l] -> [H, >=hi] in score_dtype;  acc_src_ptrs[l] -> fp32 [H, >=L] (stride acc_src_sh), acc_dst_ptrs[l] ->
 *   fp32 [H, >=L'] (stride acc_dst_sh; rows past L' are left as the caller initialised them) — both NULL to skip.
 * Everything else as spatten_prune_layers (which is this call with score_dtype == kv_dtype and no accumulators). */
int spatten_prune_layers_scored(int score_dtype, int kv_dtype, int layers,
                                const void* const* score_ptrs, int64_t score_sh,
                                const void* const* k_src_ptrs, const void* const* v_src_ptrs, int64_t src_sb, int64_t src_sh,
                                void* const* k_dst_ptrs, void* const* v_dst_ptrs, void* const* kr_dst_ptrs /* optional */,
                                int64_t dst_sb, int64_t dst_sh,
                                const void* cos, const void* sin, int table_rows, int32_t* idx,
                                const float* const* acc_src_ptrs, int64_t acc_src_sh,
                                float* const* acc_dst_ptrs, int64_t acc_dst_sh,
                                int batch, int heads, int head_dim,
                                int lo, int hi, int k, int tail_lo, int tail_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SpAtten semantics with no numeric implementation in the reference ("parity unpinned": restated from the RTL
 * control flow / README, checked against oracle/spatten_oracle.py only).
 * ---------------------------------------------------------------------------------------------- */
/* Cascade (cumulative) importance, README.md:11 / trace flag if_accumulate_importance (workloads/small.csv:1):
 *   acc[h, j] += sum over batch b and query rows i of softmax(stash[b,h,i,:] + mask[b,i,:])[j]
 * stash [B,H,q,L] (strides sb, sh, sq; L contiguous), lse [B,H,q,2] fp32 = (row max, sum exp) of the masked logits as
 * produced by spatten_attn_decode or spatten_row_lse, mask optional [B,q,L] (strides mask_sb, mask_sq),
 * acc [H, L] fp32 (stride acc_sh).  causal != 0: row i only sees keys j <= L - q + i. */
int spatten_importance_accumulate(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq,
                                  const float* lse, const void* mask, int64_t mask_sb, int64_t mask_sq,
                                  float* acc, int64_t acc_sh, int batch, int heads, int q_len, int kv_len,
                                  int causal, void* stream);
/* The same accumulation for a multi-token forward WITHOUT the [B,H,q,N] stash (4 GiB per layer at q = N = 8192): the
 * logits are recomputed on the matrix cores from q (un-rotated, rotated here like spatten_attn_prefill does) and the
 * rotated shadow, with the flash kernel's roundings, and turned into probabilities with the row statistics `lse`
 * [B,H,q_len,2] that spatten_attn_prefill wrote for the same inputs; causal != 0: the HF rule.  bf16 / f16, head_dim
 * 64 / 128.  workspace: spatten_importance_prefill_workspace_bytes (the rotated queries). */
size_t spatten_importance_prefill_workspace_bytes(int batch, int heads, int head_dim, int q_len);
int spatten_importance_accumulate_prefill(int dtype, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq,
                                          const void* kr_cache, int64_t kv_sb, int64_t kv_sh,
                                          const void* cos, const void* sin, int table_rows,
                                          const int64_t* position_ids, int64_t pos_sb,
                                          const float* lse, float* acc, int64_t acc_sh, void* workspace,
                                          int batch, int heads, int kv_heads, int head_dim,
                                          int q_len, int kv_len, int pos_q0, int causal, void* stream);
/* (row max, sum exp) of every masked logit row: lse [B,H,q,2] fp32 — for stashes that did not come with one. */
int spatten_row_lse(int dtype, const void* stash, int64_t sb, int64_t sh, int64_t sq, const void* mask,
                    int64_t mask_sb, int64_t mask_sq, float* lse, int batch, int heads, int q_len, int kv_len,
                    int causal, void* stream);
/* The accumulator follows the cache through a prune: dst[h, :] = cat(src[h, :start], src[h, idx[h, :]], src[h, tail_lo:tail_lo+tail_len]) */
int spatten_importance_compact(const float* src, int64_t src_sh, float* dst, int64_t dst_sh, const int32_t* idx,
                               int64_t idx_sh, int heads, int start, int k, int tail_lo, int tail_len, void* stream);
/* Layer-to-layer cascade (README.md:11; trace columns if_topk / topk; the survivors of a layer's top-k feed the next
 * layer): rank[h, j] = score[h, j] if the token held by slot j of THIS layer (ids[h, j]) is among the tokens the previous
 * layer kept for head h (prev_ids[h, 0..n_prev), ascending), else -inf.  spatten_topk_select over `rank` (fp32) then
 * prefers the previous layer's survivors.  score [H, >=len] model dtype; ids / prev_ids int32; rank fp32 [H, >=len]. */
int spatten_cascade_rank(int dtype, const void* score, int64_t score_sh, const int32_t* ids, int64_t ids_sh,
                         const int32_t* prev_ids, int64_t prev_sh, int n_prev, float* rank, int64_t rank_sh,
                         int heads, int len, void* stream);
/* The whole prune event of the layer-to-layer cascade in three launches (one workgroup per head walks the layers — the
 * dependency "layer l chooses among the tokens layer l-1 kept" runs per head —, one ragged K/V gather + shadow, one ragged
 * accumulator gather).  Per-layer geometry travels as a table of 16 int64 words per layer, given twice: in DEVICE memory
 * (read by the kernels) and in host memory (validated, sizes the grids):
 *   { len, hi, k, new_len,  score_sh, n_known, known_sh, new_ids_sh,  src_sb, src_sh, dst_sb, dst_sh,  acc_src_sh, acc_dst_sh,
 *     id_base, 0 }   with new_len = start + k + (len - hi), k non-increasing over the layers.
 * score_ptrs[l] -> [H, >= len] importance in score_dtype; known_ptrs[l] -> int32 [H, n_known] token ids of the slots the
 * last prune left (NULL when n_known = 0), slots j >= n_known hold token id_base + (j - n_known) (appended since);
 * new_ids_ptrs[l] -> int32 [H, new_len] out; idx int32 [layers, H, kmax] out (kept window positions, ascending);
 * key_scratch uint32 [H, key_scratch_sh >= max window]; K / V / shadow / accumulator pointer tables as in
 * spatten_prune_layers_scored (accumulators optional).  Same selection as spatten_cascade_rank + spatten_topk_select.
 * Round 4: the event is issued as a few legs of consecutive layers; each leg's gathers run on a LIBRARY-OWNED side stream
 * (one per device, created on first use) behind an event, under the next leg's selection chain, and `stream` waits for that
 * side stream before the call's work counts as done on it: for the caller everything is still ordered on `stream` (a
 * stream capture follows the fork / join).  SPATTEN_LC_LEGS=1 keeps every launch on `stream`. */
int spatten_prune_layer_cascade(int score_dtype, int kv_dtype, int layers, const void* lay_dev, const void* lay_host,
                                const void* const* score_ptrs, const int32_t* const* known_ptrs, int32_t* const* new_ids_ptrs,
                                const void* const* k_src_ptrs, const void* const* v_src_ptrs,
                                void* const* k_dst_ptrs, void* const* v_dst_ptrs, void* const* kr_dst_ptrs /* optional */,
                                const void* cos, const void* sin, int table_rows,
                                int32_t* idx, int kmax, uint32_t* key_scratch, int64_t key_scratch_sh,
                                const float* const* acc_src_ptrs, float* const* acc_dst_ptrs /* both or neither */,
                                int batch, int heads, int head_dim, int start, void* stream);
/* Head importance (head pruning, README.md:21): scores[h] += sum over b, i, d of |out[b, i, h*d : (h+1)*d]|; out [B,q,H*d]. */
int spatten_head_scores(int dtype, const void* out, int64_t out_sb, int64_t out_sq, float* scores,
                        int batch, int q_len, int heads, int head_dim, void* stream);
/* Local V pruning, pass 2 (SpAttenController.scala:546-558,591-612): out[b, h*d:(h+1)*d] = sum over the kept keys
 * j = idx[b*H+h, i] of exp(stash[b,h,j] (+mask[b,j]) - lse_max) / lse_sum * V[b, hkv, j, :]   (no renormalisation).
 * idx int32 [B*H, k] (stride idx_sr) from spatten_topk_select over the stash rows.
 * workspace (zero-filled once, spatten_pv_gather_workspace_bytes; re-armed by the kernel; one per stream): the kept
 * list is split over up to 64 workgroups per (b, h) whose partial sums the last arriver adds in split order; NULL =
 * one workgroup per (b, h). */
size_t spatten_pv_gather_workspace_bytes(int batch, int heads, int head_dim);
int spatten_pv_gather(int dtype, const void* stash, int64_t sc_sb, int64_t sc_sh, const float* lse,
                      const void* mask, int64_t mask_sb, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                      const int32_t* idx, int64_t idx_sr, int k, void* out, int64_t out_sb,
                      int batch, int heads, int kv_heads, int head_dim, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Local value pruning as ONE launch (ABI 4; SpAttenController.scala:546-558,591-612; parity unpinned): per head the `keep`
 * most probable keys of the step (k-th largest logit, ties lowest index first — TopK.scala:193-212) fetch their V row, P.V runs
 * over them with the full softmax denominator (no renormalisation).  The launch streams the rotated keys once, writes the
 * stash (`scores`, required: [B,H,>=kv_len], the logits with the reference's roundings, modify_llama.py:111-119) and (max, sum)
 * (`lse`, optional [B,H,2]), selects exactly (radix select over the head's splits, hand-overs inside the launch) and gathers only
 * the kept V rows.  It appends nothing.  Equivalent to spatten_attn_decode_args(SCORES_ONLY) + spatten_topk_select +
 * spatten_pv_gather (same stash, same kept set).
 *   keep            kept keys per head (host value), clamped to [1, kv_len]
 *   step_state      device-resident length (see "Device-resident step state"): kv_len is then the BOUND and the kept count is
 *                   ceil(keep_fraction * length) evaluated on the device in fp64 — what Python's math.ceil(f * n) gives
 *   kv_len_layout   as in spatten_decode_args_t
 *   workspace       spatten_local_v_workspace_bytes(batch, heads) bytes, zero-filled once, one per stream
 * SPATTEN_ERR_UNSUPPORTED: a split longer than 16384 rows (B*H > 256 at very long contexts): use the three calls.
 * ---------------------------------------------------------------------------------------------- */
size_t spatten_local_v_workspace_bytes(int batch, int heads);
int spatten_attn_decode_local_v(int dtype, const void* q, int64_t q_sb, int64_t q_sh, const void* kr_cache,
                                const void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos, const void* sin,
                                int table_rows, int pos_q, void* out, int64_t out_sb, void* scores, int64_t sc_sb,
                                int64_t sc_sh, float* lse, void* workspace, int batch, int heads, int kv_heads,
                                int head_dim, int kv_len, int keep, double keep_fraction, int kv_len_layout,
                                const void* step_state, void* stream);
/* The same step WITH the append inside the launch (round 5; modify_llama.py:95-104): row kv_len - 1 (step_state: state word 0 - 1)
 * of k_cache (optional) / kr_cache / v_cache <- k_new / v_new [B,Hkv,d] (strides new_sb, new_sh), the key rotated at that slot; the
 * split that owns the row scores it from registers as one extra key, the kept-row gather may read its V row back.  kv_len counts
 * the appended row.  Replaces spatten_kv_append[_step] + spatten_attn_decode_local_v: same stash, same kept set, same cache rows
 * (the head's (max, sum) and the output agree to rounding: the extra key is folded first instead of last). */
int spatten_attn_decode_local_v_append(int dtype, const void* q, int64_t q_sb, int64_t q_sh, const void* k_new,
                                       const void* v_new, int64_t new_sb, int64_t new_sh, void* k_cache, void* kr_cache,
                                       void* v_cache, int64_t kv_sb, int64_t kv_sh, const void* cos, const void* sin,
                                       int table_rows, int pos_q, void* out, int64_t out_sb, void* scores, int64_t sc_sb,
                                       int64_t sc_sh, float* lse, void* workspace, int batch, int heads, int kv_heads,
                                       int head_dim, int kv_len, int keep, double keep_fraction, int kv_len_layout,
                                       const void* step_state, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Progressive quantisation of the (rotated) key cache — MSB-first fetch with LSB refetch on low confidence
 * (MatrixFetcher.scala:48-51,341-348; RequantDecision.scala:44-72; SpAttenController.scala:35-39,402).  Parity unpinned.
 *   msb, lsb  [B,Hkv,cap,d/2] bytes: two 4-bit fields per byte (element 2i low nibble), msb signed, lsb unsigned,
 *             q8 = msb*16 + lsb;  scale [B,Hkv,cap] fp32 per row;  x ~ q8 * scale   (plane strides pl_sb, pl_sh in
 *             bytes; scale strides sc_sb, sc_sh in elements)
 * ---------------------------------------------------------------------------------------------- */
/* quantise rows [row_lo, row_hi) of the rotated shadow kr_cache into the planes */
int spatten_pq_pack(int dtype, const void* kr_cache, int64_t kv_sb, int64_t kv_sh, void* msb, void* lsb, float* scale,
                    int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh, int batch, int kv_heads, int head_dim,
                    int row_lo, int row_hi, void* stream);
/* The decode step over the planes is spatten_attn_decode_args with the pq_* fields set. */

/* ------------------------------------------------------------------------------------------------
 * Bit profiles and the quantised VALUE plane (ABI 4).  The accelerator fetches K at `profile_key.bit_count` MSBs — 4, 6
 * (two lines fused per transaction) or 8 (MatrixFetcher.scala:48-51; TestSpAtten.scala:173-176) — left-aligned in its
 * 12-bit lane, refetches 4 LSBs on low confidence (requantBitCount, SpAttenController.scala:35-39; write mask 0x00F,
 * :230-232), and fetches V ONCE at `profile_val.bit_count` bits — 8 by default, 6 in the per8 trace
 * (TestSpAtten.scala:64,83-97; SpAttenController.scala:716-723: `high_bits := True` = one fetch, not full precision).
 * Here: K is quantised symmetric per row to T = key_msb_bits + 4 bits, q = clip(rint(x / scale)), scale = amax / (2^(T-1) - 1);
 * the MSB plane holds q >> 4 (key_msb_bits signed bits), the LSB plane q & 15; pass 1 scores with 16 * msb, the refetch
 * pass adds the LSB term to the MSB logit pass 1 left in `msb_logit` (fp32: the row is recomputed ONCE, :402, without
 * re-reading the MSB plane).  V is quantised symmetric per row to value_bits: v ~ val_scale * qv.  Parity unpinned.
 *
 * Plane layouts (d = head_dim, LPR = d/16; "piece c" of a row = the 16 elements lane c of a row's LPR lanes owns in the
 * decode kernels: t < 8 -> element 8c + t, t >= 8 -> element d/2 + 8c + (t - 8), c in [0, LPR)):
 *   key_msb   [B,Hkv,cap,d*key_msb_bits/8] bytes, piece c at byte 2*key_msb_bits*c, field t (key_msb_bits wide) at bit
 *             key_msb_bits*t of the piece (little endian) = msb + 2^(key_msb_bits-1)   (NOT the spatten_pq_pack layout)
 *   key_lsb   [B,Hkv,cap,d/2] bytes, piece c at byte 8c, nibble t at bit 4t
 *   val_q     [B,Hkv,cap,d*value_bits/8] bytes, the same piece layout, field = qv + 2^(value_bits-1)
 *   key_scale, val_scale [B,Hkv,cap] fp32;  msb_logit [B,H,cap] fp32 scratch (written by pass 1, read by the refetch pass)
 * Supported profiles (key_msb_bits, value_bits): (4, 8), (8, 8) — the RTL harness default —, (6, 6) — the per8 trace.
 * (4 with V in the model dtype is the pq_* mode of spatten_attn_decode_args.)  All dtypes, head_dim 64 / 128.
 * ---------------------------------------------------------------------------------------------- */
typedef struct spatten_pq_planes {
  uint32_t struct_size;            /* sizeof(spatten_pq_planes_t) of the caller */
  int32_t key_msb_bits;            /* 4, 6 or 8 */
  int32_t value_bits;              /* 8 or 6 */
  int32_t pad0_;
  void* key_msb; void* key_lsb; float* key_scale;
  void* val_q; float* val_scale;
  float* msb_logit;
  int64_t km_sb, km_sh;            /* key_msb strides, BYTES */
  int64_t kl_sb, kl_sh;            /* key_lsb strides, BYTES */
  int64_t vq_sb, vq_sh;            /* val_q strides, BYTES */
  int64_t sc_sb, sc_sh;            /* key_scale and val_scale strides, elements */
  int64_t lg_sb, lg_sh;            /* msb_logit strides, elements */
} spatten_pq_planes_t;
/* bytes of one plane row: head_dim * bits / 8 */
size_t spatten_pq_plane_row_bytes(int head_dim, int bits);
/* Quantise rows [row_lo, row_hi) of the rotated shadow kr_cache into the key planes and of v_cache into the value plane.
 * With `step_state` (device-resident length, see above) the ONE row (state word 0) - 1 is packed instead — the step's
 * row, after spatten_kv_append_step wrote it — and row_hi is the planes' capacity (a row >= row_hi is not written). */
int spatten_pq_pack_planes(int dtype, const void* kr_cache, const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                           const spatten_pq_planes_t* planes, int batch, int kv_heads, int head_dim, int row_lo, int row_hi,
                           const void* step_state, void* stream);
/* ONE decode step's append and its plane rows in ONE launch (round 5; replaces spatten_kv_append[_step] followed by
 * spatten_pq_pack_planes for the step's row — bit for bit the same planes and cache rows; modify_llama.py:95-104 + the packing
 * above): row `row` of k_cache (optional) / v_cache <- k_new / v_new [B,Hkv,d] (strides new_sb, new_sh), of kr_cache <- k_new
 * rotated at slot position `row` (cos / sin [table_rows, d/2]), and that row of every plane.  With `step_state` the row is
 * (state word 0) - 1 and the rotary row is the state's staged one (cos / sin / row ignored): capturable.  A row >= capacity is
 * not written (step form) / refused (host form).  head_dim 64 / 128, the three bit profiles. */
int spatten_kv_append_planes(int dtype, const void* k_new, const void* v_new, int64_t new_sb, int64_t new_sh, void* k_cache,
                             void* kr_cache, void* v_cache, int64_t kv_sb, int64_t kv_sh, const spatten_pq_planes_t* planes,
                             const void* cos, const void* sin, int table_rows, int batch, int kv_heads, int head_dim, int row,
                             int capacity, const void* step_state, void* stream);
/* The decode step over the planes: pass 1 (MSB logits -> stash + msb_logit, softmax, P.V over the quantised V,
 * need_lsb[b*H+h] = max_j prob_j < threshold; RequantDecision.scala:44-72) and the refetch pass for the flagged heads (LSB plane +
 * msb_logit; V again, because the probabilities change).  Arguments as in spatten_decode_args_t; the launch appends nothing.
 * flags bit 0: pass 1 only (measurements). */
#define SPATTEN_PQ_MSB_PASS_ONLY 1
typedef struct spatten_pq_decode_args {
  uint32_t struct_size; int32_t dtype;
  const void* q; int64_t q_sb, q_sh;
  const spatten_pq_planes_t* planes;       /* host pointer */
  const void* cos; const void* sin; int32_t table_rows; int32_t pos_q;
  void* out; int64_t out_sb;
  void* scores; int64_t sc_sb, sc_sh;      /* optional stash [B,H,>=kv_len] */
  float* lse;                              /* optional [B,H,2] */
  int32_t* need_lsb; float threshold; int32_t flags;
  void* workspace; int32_t workspace_splits;
  int32_t batch, heads, kv_heads, head_dim, kv_len, n_splits, kv_len_layout;
  const int32_t* head_ids; int32_t n_active_heads; int32_t pad0_;
  float* head_abs_acc;
  const void* step_state;
  /* round 5 — the step's APPEND inside the MSB pass (all NULL / 0: the launch appends nothing, as before): row kv_len - 1
     (step_state: state word 0 - 1) of k_cache (optional) / kr_cache / v_cache [B,Hkv,cap,d] (strides kv_sb, kv_sh) <- k_new / v_new
     [B,Hkv,d] (strides new_sb, new_sh), the key rotated at that slot (modify_llama.py:95-104), and that row of every plane — what
     spatten_kv_append_planes leaves, bit for bit; the owning split scores the row from its registers.  kv_len counts the row.
     Only the LAUNCHED heads append (a head list skips pruned heads, as spatten_attn_decode_args does). */
  const void* k_new; const void* v_new; int64_t new_sb, new_sh;
  void* k_cache; void* kr_cache; void* v_cache; int64_t kv_sb, kv_sh;
} spatten_pq_decode_args_t;
int spatten_attn_decode_pq(const spatten_pq_decode_args_t* args, void* stream);

/* Prefill over the planes (BASELINE.json configs[3]: prefill + progressive quantisation).  Same query / value / output /
 * mask / position arguments as spatten_attn_prefill; the keys come from the MSB / LSB planes: pass 1 scores every query
 * row from the MSB plane (logits in fp32), need_lsb[b,h,i] = (max_j prob_ij < threshold) per QUERY ROW (int32 [B,H,q_len],
 * contiguous, written by the call), pass 2 refetches the LSB plane and recomputes the flagged rows once
 * (RequantDecision.scala:44-72, SpAttenController.scala:402).  bf16 / f16, head_dim 64 / 128 (the MFMA flash kernel). */
size_t spatten_prefill_pq_workspace_bytes(int dtype, int batch, int heads, int kv_heads, int head_dim,
                                          int q_len, int kv_len);
int spatten_attn_prefill_pq(int dtype,
                            const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sq,
                            const void* msb, const void* lsb, const float* scale,
                            int64_t pl_sb, int64_t pl_sh, int64_t sc_sb, int64_t sc_sh,
                            const void* v_cache, int64_t kv_sb, int64_t kv_sh,
                            const void* cos, const void* sin, int table_rows,
                            const int64_t* position_ids, int64_t pos_sb,
                            const void* mask, int64_t mask_sb, int64_t mask_sq,
                            void* out, int64_t out_sb, int64_t out_sq,
                            int32_t* need_lsb, float threshold, void* workspace,
                            int batch, int heads, int kv_heads, int head_dim,
                            int q_len, int kv_len, int pos_q0, int causal,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Head-parallel exchange: rank r of G owns heads [r*H/G, (r+1)*H/G) and their KV planes; token pruning
 * needs no communication; the one exchange on the path is the all-gather of the attention outputs [B, q, H/G*d] in
 * front of o_proj (and, for head pruning, of H/G fp32 head scores).  The library owns an RCCL communicator so that the
 * collective is a plain stream operation — capturable into the per-token HIP graph — and usable from any host language:
 *   rank 0: spatten_comm_unique_id(id) ; ship the 128 bytes to every rank out of band (torch.distributed, MPI, a file)
 *   every rank, after hipSetDevice: spatten_comm_init(&comm, rank, nranks, id)          (collective, blocks)
 *   per step: spatten_allgather(comm, send, recv, bytes_per_rank, stream)  -> recv[r*bytes .. ) = rank r's send buffer
 * RCCL is dlopen-ed on first use (librccl.so by name, or the path in the environment variable SPATTEN_RCCL_LIB);
 * SPATTEN_ERR_UNSUPPORTED when it cannot be loaded.  SPATTEN_ERR_INVALID: rank outside [0, nranks), NULL arguments;
 * SPATTEN_ERR_LAUNCH: RCCL refused (e.g. two ranks of one communicator on the same device).
 * ---------------------------------------------------------------------------------------------- */
#define SPATTEN_COMM_ID_BYTES 128
int spatten_comm_unique_id(void* id_out /* SPATTEN_COMM_ID_BYTES */);
int spatten_comm_init(void** comm_out, int rank, int nranks, const void* unique_id);
int spatten_comm_destroy(void* comm);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank) */
int spatten_comm_info(void* comm, int* nranks_out, int* rank_out);
int spatten_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);

/* Peer-store all-gather (ABI 4; SURVEY 8e: the decode step's exchange is latency-bound — 1 KiB per rank per layer at 8 GPUs —
 * so "single-shot direct writes to all peers, all links concurrently" instead of a collective library's ring / tree steps).
 * Every rank owns a receive window in its own HBM, mapped into every peer through hipIpc; ONE launch per all-gather writes
 * this rank's slice into slot `rank` of every peer's window, publishes an epoch flag there, waits for the peers' flags in its
 * own window and copies their slices out: recv[r*bytes .. ) = rank r's send.  A stream operation (capturable: the epoch lives
 * in device memory).  Messages up to max_bytes_per_rank (a multiple of 8 bytes); larger exchanges (prefill) use spatten_allgather.
 *   every rank:  spatten_peer_create(&peer, rank, nranks, max_bytes, my_handle)      my_handle: SPATTEN_PEER_HANDLE_BYTES
 *   out of band: all-gather the handles (rank-major);   every rank: spatten_peer_connect(peer, all_handles)
 *   per step:    spatten_peer_allgather(peer, send, recv, bytes, stream);   spatten_peer_status(peer, stream) at sync points
 * (SPATTEN_ERR_TIMEOUT: a peer's flag did not arrive within the bounded wait — its slice was filled with 0xFF).
 * nranks <= 16.  One process per GPU; HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only offers dmabuf IPC. */
#define SPATTEN_PEER_HANDLE_BYTES 64
int spatten_peer_create(void** peer_out, int rank, int nranks, size_t max_bytes_per_rank, void* handle_out);
int spatten_peer_connect(void* peer, const void* handles);
int spatten_peer_allgather(void* peer, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int spatten_peer_status(void* peer, void* stream);
int spatten_peer_destroy(void* peer);

#ifdef __cplusplus
}
#endif
#endif /* SPATTEN_H_ */
