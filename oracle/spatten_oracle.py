"""CPU oracle for the SpAtten cascade-pruned attention hot path (numpy).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``spatten_amd``) never routes through this module; it
fails loudly when the HIP library is missing.

It is a plain-numpy restatement of the reference's algorithm for the path
(every function cites the reference ``file:line`` it follows; paths are
relative to ``/root/reference``).  It is pinned against golden vectors captured
from the imported reference (``tests/golden/gen_golden.py`` ->
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Pinned by goldens      : rope table / apply_rotary_pos_emb_single, attention core
                         (decode + prefill, fp32/bf16/fp16), importance, window
                         top-k, KV compaction, caller protocol (L-trajectory).
PARITY UNPINNED (the reference has no numeric implementation; restated from the
RTL control flow / README, self-consistency tests only): cascade importance,
local V pruning, head pruning, progressive quantisation.

dtype emulation: tensors are carried as float32 arrays whose values are exactly
representable in the emulated model dtype ("f32" | "f16" | "bf16"); every torch
op of the reference that rounds to the model dtype is followed here by
``round_dt``.  torch computes 16-bit elementwise ops in fp32 (opmath) and rounds
once, which is what ``round_dt(op_in_f32)`` reproduces.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

DTYPES = ("f32", "f16", "bf16")


# --------------------------------------------------------------------------- #
# dtype emulation
# --------------------------------------------------------------------------- #
def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 -> float32 (NaN preserved)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    nan = np.isnan(x)
    if nan.any():
        rounded = np.where(nan, u | np.uint32(0x00400000), rounded) & np.uint32(0xFFFF0000)
    return rounded.view(np.float32)


def round_dt(x: np.ndarray, dt: str) -> np.ndarray:
    if dt == "f32":
        return np.asarray(x, dtype=np.float32)
    if dt == "f16":
        with np.errstate(over="ignore"):
            return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if dt == "bf16":
        return round_bf16(x)
    raise ValueError(f"unknown dtype {dt!r}")


def finfo_min(dt: str) -> float:
    """torch.finfo(dtype).min — what HF 4.33 puts above the causal diagonal."""
    return {"f32": -3.4028234663852886e38, "f16": -65504.0, "bf16": -3.3895313892515355e38}[dt]


# --------------------------------------------------------------------------- #
# rotary tables (transformers==4.33.0 LlamaRotaryEmbedding, restated; the
# reference calls it at spatten_llm/pos_shift/modify_llama.py:89)
# --------------------------------------------------------------------------- #
def rope_table(seq_len: int, dim: int, dt: str, base: float = 10000.0) -> Tuple[np.ndarray, np.ndarray]:
    """cos, sin of shape [seq_len, dim] in the model dtype.

    inv_freq = 1/base^(arange(0,dim,2)/dim) ; freqs = outer(arange(n), inv_freq)
    (both fp32) ; emb = cat(freqs, freqs) ; cos/sin in fp32 then ``.to(x.dtype)``.
    """
    # torch's fp32 pow is correctly rounded here: pow in fp64, round to fp32, then the fp32 reciprocal
    # reproduces torch's inv_freq bit-for-bit (checked for dim 64 and 128 in gen_golden's container).
    expo = (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)).astype(np.float64)
    inv_freq = (np.float32(1.0) / (float(base) ** expo).astype(np.float32)).astype(np.float32)
    t = np.arange(seq_len, dtype=np.float32)
    freqs = (t[:, None] * inv_freq[None, :]).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1).astype(np.float64)
    # cos/sin of the fp32 angle, correctly rounded to fp32 (torch's differs by <= 1 fp32 ulp)
    return round_dt(np.cos(emb).astype(np.float32), dt), round_dt(np.sin(emb).astype(np.float32), dt)


def rotate_half(x: np.ndarray) -> np.ndarray:
    """transformers rotate_half: cat(-x[d/2:], x[:d/2]) (modify_llama.py:12,27)."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rotary_pos_emb_single(x: np.ndarray, cos: np.ndarray, sin: np.ndarray,
                                position_ids: np.ndarray, dt: str) -> np.ndarray:
    """modify_llama.py:21-28.  x [B,H,n,d]; cos/sin [S,d]; position_ids [B,n] (or [1,n]).

    ``(x*cos) + (rotate_half(x)*sin)`` — three ops, each rounded to the model dtype.
    """
    c = cos[position_ids][:, None, :, :]  # [B,1,n,d]
    s = sin[position_ids][:, None, :, :]
    a = round_dt(x * c, dt)
    b = round_dt(rotate_half(x) * s, dt)
    return round_dt(a + b, dt)


# --------------------------------------------------------------------------- #
# attention core (modify_llama.py:86-147 )
# --------------------------------------------------------------------------- #
def repeat_kv(x: np.ndarray, n_rep: int) -> np.ndarray:
    """transformers repeat_kv (modify_llama.py:108-109)."""
    if n_rep == 1:
        return x
    return np.repeat(x, n_rep, axis=1)


def attention_core(q: np.ndarray, k_new: np.ndarray, v_new: np.ndarray,
                   past_k: Optional[np.ndarray], past_v: Optional[np.ndarray],
                   position_ids: np.ndarray, mask: Optional[np.ndarray], dt: str,
                   base: float = 10000.0):
    """The attention core between the q/k/v projections and o_proj.

    q [B,H,q,d]; k_new/v_new [B,Hkv,q,d] un-rotated; past_k/past_v [B,Hkv,P,d]
    (K cached UN-rotated, modify_llama.py:100); position_ids [B,q] int;
    mask additive [B,1,q,N] or None.

    Returns (attn_output [B,q,H*d], attn_scores stash [B,H,q,N] (pre-mask,
    pre-softmax, modify_llama.py:116-119), (K_cache, V_cache)).
    """
    B, H, ql, d = q.shape
    Hkv = k_new.shape[1]
    P = 0 if past_k is None else past_k.shape[2]
    N = P + ql
    cos, sin = rope_table(N, d, dt, base)                                   # :89
    qr = apply_rotary_pos_emb_single(q, cos, sin, position_ids, dt)        # :92
    kc = k_new if past_k is None else np.concatenate([past_k, k_new], 2)   # :95-98
    vc = v_new if past_v is None else np.concatenate([past_v, v_new], 2)
    key_pos = np.arange(N)[None, :]                                        # :103
    kr = apply_rotary_pos_emb_single(kc, cos, sin, key_pos, dt)            # :104
    kr = repeat_kv(kr, H // Hkv)                                           # :108
    vr = repeat_kv(vc, H // Hkv)                                           # :109
    s = round_dt(np.matmul(qr, np.swapaxes(kr, 2, 3)), dt)                 # :111 matmul (fp32 acc) -> dtype
    s = round_dt(s / np.float32(math.sqrt(d)), dt)                         # :111-113 separate divide
    stash = s.copy()                                                       # :116-119
    if mask is not None:
        s = round_dt(s + mask, dt)                                         # :132
    s32 = s.astype(np.float32)                                             # :135 softmax in fp32
    m = s32.max(axis=-1, keepdims=True)
    e = np.exp(s32 - m).astype(np.float32)
    p = round_dt(e / e.sum(axis=-1, keepdims=True, dtype=np.float32), dt)  # :135-137 .to(dtype)
    o = round_dt(np.matmul(p, vr), dt)                                     # :138
    o = np.swapaxes(o, 1, 2).reshape(B, ql, H * d)                         # :146-147
    return o, stash, (kc, vc)


def causal_mask(B: int, ql: int, N: int, dt: str) -> np.ndarray:
    """HF 4.33 additive causal mask [B,1,q,N]: 0 for j <= P+i, finfo.min otherwise."""
    P = N - ql
    i = np.arange(ql)[:, None]
    j = np.arange(N)[None, :]
    m = np.where(j <= P + i, np.float32(0.0), np.float32(finfo_min(dt))).astype(np.float32)
    return np.broadcast_to(m[None, None], (B, 1, ql, N)).copy()


# --------------------------------------------------------------------------- #
# prune event (kv_cache_token_pruning.py:42-96 )
# --------------------------------------------------------------------------- #
def importance(stash: np.ndarray, dt: str) -> np.ndarray:
    """kv_cache_token_pruning.py:51 — ``item.sum(0).sum(1)``: [B,H,q,L] -> [H,L].

    Two reductions, each accumulated in fp32 and rounded to the stash dtype.
    """
    s0 = round_dt(stash.astype(np.float32).sum(axis=0, dtype=np.float32), dt)
    return round_dt(s0.sum(axis=1, dtype=np.float32), dt)


def ordered_key(x: np.ndarray) -> np.ndarray:
    """Monotone float32 -> uint32 key (larger value => larger key; NaN largest,
    -0.0 == +0.0), the total order torch.topk(largest=True) ranks by."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    x = np.where(x == 0.0, np.float32(0.0), x)
    u = x.view(np.uint32)
    key = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000))
    return np.where(np.isnan(x), np.uint32(0xFFFFFFFF), key).astype(np.uint32)


def topk_window(score: np.ndarray, lo: int, hi: int, k: int) -> np.ndarray:
    """kv_cache_token_pruning.py:59-63 — per head: top-k of score[:, lo:hi], indices
    sorted ascending, + lo.  Python-slice semantics for ``hi`` (clamps to L).

    Tie policy at the k-th value (torch's order there is implementation defined,
    SURVEY §7.2): keep every value > threshold, then the LOWEST-index elements
    equal to the threshold (the RTL's rule, TopK.scala:193-212).
    Returns int32 [H,k].
    """
    H, L = score.shape
    hi = min(hi, L)
    if lo < 0 or hi - lo < k or k <= 0:
        raise ValueError(f"top-k window [{lo},{hi}) holds fewer than k={k} candidates")
    out = np.empty((H, k), dtype=np.int32)
    for h in range(H):
        key = ordered_key(score[h, lo:hi])
        # stable sort of descending key == (value desc, index asc)
        order = np.argsort(~key, kind="stable")[:k]
        out[h] = np.sort(order).astype(np.int32) + lo
    return out


def kv_compact(K: np.ndarray, V: np.ndarray, idx: np.ndarray, start: int, tail_lo: int):
    """kv_cache_token_pruning.py:64-96 — mask-gather of the kept rows and the
    concat [start | important | recent'] for K and V.  K,V [B,H,L,d]; idx [H,k]
    absolute positions.  ``tail_lo`` = L - recent + num_coming (tail empty if >= L)."""
    B, H, L, d = K.shape
    tail_lo = min(max(tail_lo, 0), L)
    outs = []
    for X in (K, V):
        imp = np.stack([X[:, h, idx[h], :] for h in range(H)], axis=1)
        outs.append(np.concatenate([X[:, :, :start], imp, X[:, :, tail_lo:L]], axis=2))
    return outs[0], outs[1]


def apply_token_pruning(past, num_coming: int, stash_all: Sequence[np.ndarray],
                        start: int, recent: int, important_size: int, dt: str):
    """SpAttenKVCache.apply_token_pruning (kv_cache_token_pruning.py:42-96).

    past: list over layers of (K,V) [B,H,L,d].  Returns (new_past, idx_per_layer);
    ``new_past is past`` in the passthrough case, None for None.
    """
    if past is None:
        return None, None
    L = past[0][0].shape[2]
    if L + num_coming <= start + important_size + recent:                     # :46-47
        return past, None
    if important_size <= 0:
        raise ValueError("important_size must be > 0 (reference raises TypeError at :63)")
    lo, hi = start, L - recent + num_coming                                   # :59
    new_past, idxs = [], []
    for (K, V), stash in zip(past, stash_all):
        score = importance(stash, dt)                                         # :51
        idx = topk_window(score, lo, hi, important_size)                      # :59-63
        Kn, Vn = kv_compact(K, V, idx, start, hi)                             # :64-96
        new_past.append([Kn, Vn])
        idxs.append(idx)
    return new_past, idxs


# --------------------------------------------------------------------------- #
# PARITY UNPINNED restatements (no numeric implementation in the reference)
# --------------------------------------------------------------------------- #
def softmax_probs(stash: np.ndarray, mask: Optional[np.ndarray] = None) -> np.ndarray:
    s = stash.astype(np.float32) if mask is None else (stash + mask).astype(np.float32)
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def cascade_importance_accumulate(acc: np.ndarray, stash: np.ndarray,
                                  mask: Optional[np.ndarray] = None) -> np.ndarray:
    """Paper semantics (README.md:11; trace flag ``if_accumulate_importance``,
    workloads/small.csv:1): importance[h,j] += sum over batch and query rows of
    the softmax probability of key j.  acc [H,L] fp32, stash [B,H,q,L]."""
    p = softmax_probs(stash, mask)
    return (acc + p.sum(axis=(0, 2), dtype=np.float32)).astype(np.float32)


def local_value_prune(probs: np.ndarray, V: np.ndarray, keep: int) -> np.ndarray:
    """Local V pruning (SpAttenController.scala:546-558,591-612): per head keep the
    ``keep`` largest probabilities (lowest-index-first on ties), P.V over them
    only, no renormalisation (the RTL multiplies the surviving probs as they are).
    probs [B,H,L] (q=1), V [B,H,L,d] -> [B,H,d]."""
    B, H, L = probs.shape
    out = np.zeros((B, H, V.shape[-1]), dtype=np.float32)
    for b in range(B):
        idx = topk_window(probs[b], 0, L, keep) if keep < L else np.tile(np.arange(L, dtype=np.int32), (H, 1))
        for h in range(H):
            out[b, h] = probs[b, h, idx[h]].astype(np.float32) @ V[b, h, idx[h]].astype(np.float32)
    return out


def head_scores(attn_out: np.ndarray, H: int) -> np.ndarray:
    """Head importance = sum |attn_out_h| over batch, queries and d (SpAtten paper;
    README.md:21).  attn_out [B,q,H*d] -> [H] fp32."""
    B, ql, hd = attn_out.shape
    return np.abs(attn_out.astype(np.float32)).reshape(B, ql, H, hd // H).sum(axis=(0, 1, 3), dtype=np.float32)


def head_prune_select(scores: np.ndarray, keep: int) -> np.ndarray:
    """Keep the ``keep`` highest-scoring heads, lowest-index-first on ties, ascending."""
    return topk_window(scores[None, :], 0, scores.shape[0], keep)[0]


def head_prune_cascade(layer_scores: Sequence[np.ndarray], keep: Sequence[int],
                       prev_kept: Optional[Sequence[Optional[np.ndarray]]] = None) -> List[np.ndarray]:
    """Cascade head pruning at one prune event (README.md:21: heads "selected on the fly", and a head pruned in one
    layer is gone in all following layers; ranked by the same top-k engine as tokens).  Restated rule: head importance
    accumulates layer over layer (cum_l = sum_{l' <= l} scores_{l'}); layer l keeps the ``keep[l]`` best heads among
    those layer l-1 just kept and — if it pruned before — those it still had (``prev_kept[l]``); ties: lowest id.
    Returns the ascending kept ids per layer."""
    H = layer_scores[0].shape[0]
    cum = np.zeros(H, np.float32)
    alive = np.ones(H, bool)
    out = []
    for l, sc in enumerate(layer_scores):
        cum = (cum + sc.astype(np.float32)).astype(np.float32)
        if prev_kept is not None and prev_kept[l] is not None:
            mine = np.zeros(H, bool)
            mine[prev_kept[l]] = True
            alive = alive & mine
        k = min(int(keep[l]), H, int(alive.sum()))
        ids = topk_window(np.where(alive, cum, -np.inf).astype(np.float32)[None, :], 0, H, k)[0]
        alive = np.zeros(H, bool)
        alive[ids] = True
        out.append(ids)
    return out


def global_token_scores(score: np.ndarray) -> np.ndarray:
    """Global (cross-head) token importance — PARITY UNPINNED.  The SpAtten engine ranks TOKENS, not (head, token) pairs:
    README.md:21 ("top-k engine to rank token and head importance"), and the traces carry ONE ``key_fetch_num`` and one
    ``if_topk`` / ``topk`` / ``if_accumulate_importance`` row per layer for all heads (spatten_hardware/hardware/workloads/
    small.csv:1).  Restated rule: a token's importance is the sum of its per-head importance rows; every head then keeps the
    SAME set.  score [H, L] -> [H, L] with every row = float32(sum over heads taken in float64) — the wide accumulation makes
    the ranking independent of the order in which heads (or head-parallel ranks) are added."""
    g = score.astype(np.float64).sum(axis=0).astype(np.float32)
    return np.broadcast_to(g, score.shape).copy()


def global_token_prune(past, num_coming: int, scores: Sequence[np.ndarray], start: int, recent: int, important_size: int):
    """The prune event of kv_cache_token_pruning.py:42-96 with ``global_token_scores`` as the ranking: same window, same
    start / important / recent concat, one kept set per layer shared by all heads.  scores[l] [H, L] = the per-head
    importance (reference mode: ``importance(stash)``; cascade mode: the accumulators).  Returns (new_past, idx_per_layer)."""
    L = past[0][0].shape[2]
    lo, hi = start, L - recent + num_coming
    new_past, idxs = [], []
    for (K, V), sc in zip(past, scores):
        idx = topk_window(global_token_scores(sc[:, :L]), lo, hi, important_size)
        Kn, Vn = kv_compact(K, V, idx, start, hi)
        new_past.append([Kn, Vn])
        idxs.append(idx)
    return new_past, idxs


def layer_cascade_prune(past, ids, scores, num_coming: int, start: int, recent: int, keeps: Sequence[int]):
    """Layer-to-layer cascade token pruning at one prune event (README.md:11 "cascade"; the traces' per-layer
    key_fetch_num shrinks layer by layer, workloads/*.csv columns if_topk / topk; TopK.scala:113-224 ranks, the
    survivors feed the NEXT layer).  Restated rule, per head:
      * every layer keeps its first ``start`` rows and its tail ``[L_l - recent + num_coming, L_l)`` like the
        reference prune (kv_cache_token_pruning.py:72-96);
      * layer l keeps ``keeps[l]`` rows of its window ``[start, L_l - recent + num_coming)``, ranked by its own
        importance — but among the tokens layer l-1 just kept (a token pruned by a layer is gone for the layers after
        it); only when fewer than keeps[l] such tokens are in the window is it filled with the lowest-index others.
    past[l] = (K, V) [B,H,L_l,d]; ids[l] int [H, L_l] = token id held by each slot (ascending per head);
    scores[l] [H, L_l].  Returns (new_past, new_ids, idx_per_layer)."""
    new_past, new_ids, idxs = [], [], []
    prev = None
    for l, ((K, V), tid, sc) in enumerate(zip(past, ids, scores)):
        H, L = sc.shape
        lo, hi = start, min(L - recent + num_coming, L)
        rank = sc.astype(np.float32).copy()
        if prev is not None:
            for h in range(H):
                member = np.isin(tid[h], prev[h])
                rank[h] = np.where(member, rank[h], -np.inf)
        idx = topk_window(rank, lo, hi, int(keeps[l]))
        Kn, Vn = kv_compact(K, V, idx, start, hi)
        nid = np.stack([np.concatenate([tid[h, :start], tid[h, idx[h]], tid[h, hi:L]]) for h in range(H)])
        new_past.append([Kn, Vn])
        new_ids.append(nid)
        idxs.append(idx)
        prev = nid
    return new_past, new_ids, idxs


def pq_logits(qr: np.ndarray, msb, lsb, scale, threshold: float, lsb_bits: int = 4):
    """The logits a progressive-quant decode step actually uses (pass-1 MSB logits, or the refetched 8-bit logits for
    the rows whose pass-1 max probability is below ``threshold``) and the refetch flags.  qr [B,H,d]; planes [B,H,L,d].
    Returns (logits [B,H,L] fp32, need [B,H] bool)."""
    d = qr.shape[-1]
    s1 = np.einsum("bhd,bhld->bhl", qr.astype(np.float32), pq_dequant(msb, None, scale, lsb_bits)) / np.float32(math.sqrt(d))
    need = softmax_probs(s1).max(axis=-1) < np.float32(threshold)
    s2 = np.einsum("bhd,bhld->bhl", qr.astype(np.float32), pq_dequant(msb, lsb, scale, lsb_bits)) / np.float32(math.sqrt(d))
    return np.where(need[..., None], s2, s1).astype(np.float32), need


def pq_prefill_attention(qr: np.ndarray, msb, lsb, scale, V: np.ndarray, threshold: float, past_len: int,
                         rows: Optional[Sequence[int]] = None, lsb_bits: int = 4):
    """Progressive quantisation applied to a BLOCK of query rows (BASELINE.json configs[3]; the RTL takes the
    max-probability decision and the one refetch per QUERY ROW: RequantDecision.scala:44-72, SpAttenController.scala:402).
    qr [B,H,q,d] rotated queries; planes [B,H,N,d] of the rotated keys, N = past_len + q; V [B,H,N,d]; HF causal rule
    (row i sees keys j <= past_len + i).  ``rows`` restricts the computation to some query rows (large shapes).
    Returns (out [B,len(rows),H*d] fp32, need [B,H,len(rows)] bool, pmax [B,H,len(rows)])."""
    B, H, ql, d = qr.shape
    rows = list(range(ql)) if rows is None else list(rows)
    k1 = pq_dequant(msb, None, scale, lsb_bits)
    k2 = pq_dequant(msb, lsb, scale, lsb_bits)
    out = np.zeros((B, len(rows), H, V.shape[-1]), np.float32)
    need = np.zeros((B, H, len(rows)), bool)
    pmax = np.zeros((B, H, len(rows)), np.float32)
    for n, i in enumerate(rows):
        vis = past_len + i + 1
        qi = qr[:, :, i].astype(np.float32)
        s1 = np.einsum("bhd,bhld->bhl", qi, k1[:, :, :vis]) / np.float32(math.sqrt(d))
        p1 = softmax_probs(s1)
        pmax[:, :, n] = p1.max(axis=-1)
        need[:, :, n] = pmax[:, :, n] < np.float32(threshold)
        s2 = np.einsum("bhd,bhld->bhl", qi, k2[:, :, :vis]) / np.float32(math.sqrt(d))
        p = np.where(need[:, :, n, None], softmax_probs(s2), p1)
        out[:, n] = np.einsum("bhl,bhld->bhd", p, V[:, :, :vis].astype(np.float32))
    return out.reshape(B, len(rows), -1), need, pmax


def pq_quantize(K: np.ndarray, bits: int = 8, lsb_bits: int = 4):
    """Progressive-quantisation storage (MatrixFetcher.scala:48-51,341-348;
    SpAttenController.scala:35-39): symmetric per-row linear quantiser to ``bits``
    signed bits, stored as an MSB plane (bits-lsb_bits) and an LSB plane (lsb_bits).
    K [..., d] -> (msb int8 in [-2^(m-1), 2^(m-1)-1], lsb uint8 in [0, 2^l-1], scale fp32 [...,1])
    with q = msb*2^l + lsb exactly."""
    K = K.astype(np.float32)
    qmax = float(2 ** (bits - 1) - 1)
    amax = np.abs(K).max(axis=-1, keepdims=True)
    scale = np.where(amax > 0, amax / qmax, np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(K / scale), -qmax - 1, qmax).astype(np.int32)
    msb = (q >> lsb_bits).astype(np.int8)                  # arithmetic shift = floor
    lsb = (q & ((1 << lsb_bits) - 1)).astype(np.uint8)
    return msb, lsb, scale


def pq_dequant(msb: np.ndarray, lsb: Optional[np.ndarray], scale: np.ndarray, lsb_bits: int = 4) -> np.ndarray:
    """MSB-only view = MSBs left-aligned with zero LSBs (MatrixFetcher.scala:345,
    resizeLeft); full view ORs the LSB plane into the low bits (:347)."""
    q = msb.astype(np.int32) << lsb_bits
    if lsb is not None:
        q = q | lsb.astype(np.int32)
    return (q.astype(np.float32) * scale).astype(np.float32)


def pq_decode_attention(qr: np.ndarray, msb, lsb, scale, V: np.ndarray, threshold: float, lsb_bits: int = 4):
    """Progressive-quant decode (RequantDecision.scala:44-72; SpAttenController.scala:402):
    pass 1 scores from the MSB plane; ``need = max_j prob_j < threshold``; rows that
    need it refetch the LSB plane and recompute ONCE.  qr [B,H,d] (already rotated /
    keys here are taken as already position-encoded), planes [B,H,L,d], V [B,H,L,d].
    Returns (out [B,H,d] fp32, need_lsb [B,H] bool)."""
    d = qr.shape[-1]
    k1 = pq_dequant(msb, None, scale, lsb_bits)
    s1 = np.einsum("bhd,bhld->bhl", qr.astype(np.float32), k1) / np.float32(math.sqrt(d))
    p1 = softmax_probs(s1)
    need = p1.max(axis=-1) < np.float32(threshold)
    k2 = pq_dequant(msb, lsb, scale, lsb_bits)
    s2 = np.einsum("bhd,bhld->bhl", qr.astype(np.float32), k2) / np.float32(math.sqrt(d))
    p2 = softmax_probs(s2)
    p = np.where(need[..., None], p2, p1)
    return np.einsum("bhl,bhld->bhd", p, V.astype(np.float32)).astype(np.float32), need


def pq_quantize_values(V: np.ndarray, bits: int = 8):
    """The quantised VALUE plane: V is fetched ONCE at ``profile_val.bit_count`` bits — 8 by default, 6 in the per8 trace
    (TestSpAtten.scala:64,83-97,175-176; SpAttenController.scala:716-723 `high_bits := True`: one fetch, not full
    precision).  Symmetric per-row linear quantiser like the keys'.  V [..., d] -> (qv int32, vscale fp32 [..., 1])."""
    V = V.astype(np.float32)
    qmax = float(2 ** (bits - 1) - 1)
    amax = np.abs(V).max(axis=-1, keepdims=True)
    vscale = np.where(amax > 0, amax / qmax, np.float32(1.0)).astype(np.float32)
    qv = np.clip(np.rint(V / vscale), -qmax - 1, qmax).astype(np.int32)
    return qv, vscale


def pq_decode_attention_profile(qr: np.ndarray, msb, lsb, scale, qv, vscale, threshold: float, lsb_bits: int = 4):
    """Progressive-quant decode over a bit PROFILE (MatrixFetcher.scala:48-51: MSB plane of 4 / 6 / 8 bits; 4 LSBs on
    refetch, SpAttenController.scala:35-39) with the quantised value plane: pass 1 = logits from the MSB plane
    (left-aligned, MatrixFetcher.scala:345), need = max prob < threshold (RequantDecision.scala:44-72); flagged heads are
    recomputed ONCE (SpAttenController.scala:402) as logit_msb + (q . lsb) scale / sqrt(d) — the LSB plane is added to the
    pass-1 logit, the MSB plane is not fetched again (write mask 0x00F, :230-232) —; P.V over V ~ qv * vscale, P in fp32.
    qr [B,H,d] rotated queries; planes [B,H,L,d].  Returns (out [B,H,d] fp32, need [B,H] bool, logits [B,H,L] fp32)."""
    d = qr.shape[-1]
    q32 = qr.astype(np.float32)
    rs = np.float32(math.sqrt(d))
    s1 = np.einsum("bhd,bhld->bhl", q32, pq_dequant(msb, None, scale, lsb_bits)) / rs
    p1 = softmax_probs(s1)
    need = p1.max(axis=-1) < np.float32(threshold)
    s2 = s1 + np.einsum("bhd,bhld->bhl", q32, lsb.astype(np.float32) * scale) / rs
    p2 = softmax_probs(s2)
    p = np.where(need[..., None], p2, p1)
    logits = np.where(need[..., None], s2, s1).astype(np.float32)
    V = qv.astype(np.float32) * vscale
    return np.einsum("bhl,bhld->bhd", p, V).astype(np.float32), need, logits


# --------------------------------------------------------------------------- #
# synthetic inputs: counter-based generator shared by tests / bench / goldens
# --------------------------------------------------------------------------- #
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_normal(seed: int, tensor_id: int, shape, dt: str = "f32", scale: float = 1.0) -> np.ndarray:
    """g(seed, tensor_id, flat_index) -> N(0,1)*scale, rounded to ``dt``.
    splitmix64 counter -> two uniforms -> Box-Muller; independent of torch's RNG."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(tensor_id))
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + base
        a = _splitmix64(ctr)
        b = _splitmix64(ctr + np.uint64(1))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740993.0
    u2 = (b >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return round_dt((z * scale).astype(np.float32).reshape(shape), dt)
