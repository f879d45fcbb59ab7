"""SpAttenKVCache — host-side mirror of the reference's KV-cache pruning object.

Reference: spatten_llm/kv_cache_token_pruning.py:23-96.  Same constructor, attributes, method name,
argument meaning and return conventions (None -> None, passthrough returns the SAME object, otherwise
a new ``list[[K', V'], ...]``); the per-layer Python loop of torch.topk / sort / scatter / .cpu() /
boolean gather / cat is replaced by two HIP launches covering every layer (``ops.prune_layers``):
radix-select top-k that emits ascending positions directly, and one fused gather+concat of K and V.

Documented divergences (reference quirks):
  * batch > 1 and num_heads == 1 work (the reference's ``.squeeze()[mask]`` raises IndexError);
  * ``important_size == 0`` raises ValueError (reference: TypeError at :63);
  * a candidate window shorter than ``important_size`` raises ValueError (reference: torch.topk RuntimeError);
  * ties exactly at the k-th score keep the LOWEST positions (torch's order there is unspecified);
  * the returned tensors are views of slabs with spare capacity for ``num_coming`` tokens so the
    attention forward can append in place instead of re-``cat``-ing the whole cache every token.

Extension (PARITY UNPINNED): ``token_scope="global"`` — ONE kept token set per layer, shared by every head, ranked by the
importance summed over the heads (the SpAtten paper / RTL semantic: README.md:21 "top-k engine to rank token and head
importance"; the traces carry one ``key_fetch_num`` and one ``if_topk`` / ``topk`` row per layer for all heads,
spatten_hardware/hardware/workloads/small.csv:1).  The reference's Python prunes per head (kv_cache_token_pruning.py:59-69),
which stays the default (``token_scope="head"``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import kv_slab, ops


def slice2d(x, start, end):
    return x[:, :, start:end, ...]


def slice3d(x, start, end):
    return x[:, :, :, start:end, ...]


def slice1d(x, start, end):
    return x[:, start:end, ...]


DIM_TO_SLICE = {1: slice1d, 2: slice2d, 3: slice3d}


class SpAttenKVCache:
    def __init__(self, start_size=4, recent_size=128, important_size=128, k_seq_dim=2, v_seq_dim=2,
                 importance_mode="reference", token_scope="head"):
        # the reference prints this banner from its constructor (kv_cache_token_pruning.py:32)
        print(f"SpAttenKVCache: keep start: {start_size}, keep recent: {recent_size}, keep important: {important_size}")
        self.start_size = start_size
        self.recent_size = recent_size
        self.important_size = important_size
        self.cache_size = start_size + important_size + recent_size
        self.k_seq_dim = k_seq_dim
        self.v_seq_dim = v_seq_dim
        self.k_slice = DIM_TO_SLICE[k_seq_dim]
        self.v_slice = DIM_TO_SLICE[v_seq_dim]
        if importance_mode not in ("reference", "cascade"):
            raise ValueError("importance_mode must be 'reference' or 'cascade'")
        # "reference": importance = the last forward's stashed raw logits (kv_cache_token_pruning.py:51).
        # "cascade"  : importance = running sum of softmax probabilities over every forward since the key entered
        #              the cache (SpAtten paper / README.md:11; PARITY UNPINNED), accumulated by the patched forward.
        self.importance_mode = importance_mode
        if token_scope not in ("head", "global"):
            raise ValueError("token_scope must be 'head' or 'global'")
        # "head"  : every head keeps its own top-k (kv_cache_token_pruning.py:59-69).
        # "global": one kept set per layer for all heads, ranked by the importance summed over the heads (README.md:21,
        #           workloads/small.csv:1; PARITY UNPINNED; oracle: global_token_scores).  Head-parallel ranks all-reduce the
        #           [layers, L] sums once per prune event (``head_parallel``, set by enable_spatten_llm).
        self.token_scope = token_scope
        self.head_parallel = None
        self.ext = None                     # spatten_amd.extensions.SpattenExtensions (enable_spatten_llm's opt-in modes)
        self.importance_score: Optional[List[torch.Tensor]] = None
        self.keep_indices: Optional[torch.Tensor] = None      # int32 [layers, H, important] of the last prune
        self.n_pruned_last = 0
        self.n_pruned_total = 0

    def restore_process_options(self):
        """Undo what ``enable_spatten_llm(fused_step=True)`` changed process-wide (the decode team, spatten_decode_set_team)."""
        prev = getattr(self, "_prev_decode_team", None)
        if prev is not None:
            ops.set_decode_team(prev)
            self._prev_decode_team = None

    # -- pure host logic (unit-tested without a GPU) --------------------------------------------
    def window(self, seq_len: int, num_coming: int):
        """(lo, hi, tail_lo, new_len) of a prune at this length, Python-slice semantics of
        ``score[:, start : seq_len - recent + num_coming]`` (kv_cache_token_pruning.py:59, 80-82)."""
        lo = self.start_size
        hi = min(seq_len - self.recent_size + num_coming, seq_len)
        new_len = self.start_size + self.important_size + (seq_len - hi)
        return lo, hi, hi, new_len

    def needs_pruning(self, seq_len: int, num_coming: int) -> bool:
        return seq_len + num_coming > self.cache_size                         # :46

    def apply_token_pruning(self, past_key_values, num_coming, attn_score_all):
        if past_key_values is None:                                           # :43-44
            return None
        seq_len = past_key_values[0][0].size(self.k_seq_dim)
        if not self.needs_pruning(seq_len, num_coming):                       # :46-47
            return past_key_values
        if self.k_seq_dim != 2 or self.v_seq_dim != 2:
            raise NotImplementedError("the HIP path implements the llama layout [B, H, L, d] (seq dim 2)")
        if self.important_size <= 0:
            raise ValueError("important_size must be > 0 on the pruning branch")
        lo, hi, tail_lo, new_len = self.window(seq_len, num_coming)
        if hi - lo < self.important_size:
            raise ValueError(
                f"top-k window [{lo},{hi}) holds fewer than important_size={self.important_size} candidates "
                f"(seq_len={seq_len}, num_coming={num_coming})")
        n_layers = len(past_key_values)
        ops.check_workspaces()              # a natural sync point: surface a device-side merge timeout, if any
        if self.head_parallel is not None:
            self.head_parallel.peer_status()    # ... and a peer slice that never arrived (peer-store all-gather), if any
        if self.ext is not None:
            self.ext.before_prune()         # fold the pending decode step, pick the heads that survive
        if self.ext is not None and self.ext.layer_keep is not None:
            return self._prune_layer_cascade(past_key_values, num_coming, attn_score_all)
        if self.importance_mode == "cascade":
            return self._prune_cascade(past_key_values, seq_len, num_coming, lo, hi, new_len)
        if len(attn_score_all) != n_layers:
            raise ValueError("attn_score_all must hold one stash per layer")

        # importance = stash.sum(0).sum(1)  (:51) — a view when B == q == 1 (decode stash)
        self.importance_score = [ops.importance(s) for s in attn_score_all]
        for s in self.importance_score:
            if s.shape[1] < hi:
                raise ValueError("attention-score stash is shorter than the KV cache")
        kv_heads = past_key_values[0][0].shape[1]
        if self.importance_score[0].shape[0] != kv_heads:
            # grouped-query attention: one cached K/V head serves a GROUP of query heads.  The reference cannot prune such
            # a cache at all (its [H, L] mask meets a [Hkv, L, d] tensor: SURVEY A5); here a key's importance is the sum
            # of its group's rows — the rows of ``importance_score`` are then KV heads
            # (a pruned query head is not launched any more: its stash row is stale — it does not vote)
            self.importance_score = [_group_rows(self._live_rows(layer, s), kv_heads) for layer, s in enumerate(self.importance_score)]
        scores = _common_rows(self.importance_score)
        if self.token_scope == "global":
            scores = self._global_scores([s[:, :seq_len] for s in scores])
        Ks = [_rows(kv[0]) for kv in past_key_values]
        Vs = [_rows(kv[1]) for kv in past_key_values]
        Ks, Vs = _common_strides(Ks, Vs)
        B, H, _, d = Ks[0].shape
        cap = kv_slab.round_capacity(new_len + max(int(num_coming), 0))
        base, scaling = _rope_of(past_key_values)
        rope = kv_slab.rope_tables(cap, d, Ks[0].dtype, Ks[0].device, base, scaling)
        Kn, Vn, Krn, idx = ops.prune_layers(scores, Ks, Vs, seq_len, lo, hi, self.important_size,
                                            capacity=cap, rope=rope)
        self.keep_indices = idx
        self.n_pruned_last = seq_len - new_len
        self.n_pruned_total += self.n_pruned_last
        out = []
        for layer, (k, v, kr) in enumerate(zip(Kn, Vn, Krn)):
            # remember the slab (spare capacity + rotated shadow) on the tensor HF will hand back to the forward
            kv_slab.attach(k, v, kr, new_len, base, scaling)
            if self.ext is not None:
                self.ext.layers[layer].pending_len = 0
            out.append([k, v])
        return out                                                            # list of lists (:72-96)

    def _global_scores(self, scores: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """``token_scope="global"``: per layer, the importance summed over the heads — over the heads of EVERY rank when the
        cache is head-parallel (one all-reduce of the [layers, L] sums per prune event) — handed to the select as H rows of
        stride 0, so every head keeps the same set and the gather / accumulator rows downstream are unchanged.  The sum is
        taken in fp64 and rounded once to fp32: the ranking then does not depend on the order in which heads or ranks are
        added (oracle: global_token_scores)."""
        H = scores[0].shape[0]
        g = torch.stack([s.sum(0, dtype=torch.float64) for s in scores])           # [layers, L]
        hp = self.head_parallel
        if hp is not None and hp.world > 1:
            g = hp.all_reduce_sum(g)
        g = g.to(torch.float32)
        return [g[layer].unsqueeze(0).expand(H, -1) for layer in range(len(scores))]

    def _live_rows(self, layer: int, score: torch.Tensor) -> torch.Tensor:
        """Grouped-query caches under head pruning: the rows of pruned query heads zeroed before the group sum."""
        st = self.ext.layers[layer] if self.ext is not None and layer < len(self.ext.layers) else None
        pruned = None if st is None else st.pruned_ids
        if pruned is None or pruned.numel() == 0:
            return score
        return score.index_fill(0, pruned.to(score.device), 0)

    def _prune_cascade(self, past_key_values, seq_len, num_coming, lo, hi, new_len):
        """Same window / row map as the reference prune, but ranked by the accumulated probabilities; the
        accumulators are compacted with the cache so they keep describing the surviving keys."""
        if self.ext is None or any(st.acc is None for st in self.ext.layers[:len(past_key_values)]):
            raise RuntimeError("cascade importance has not been accumulated: run the patched forward first "
                               "(enable_spatten_llm(..., importance_mode='cascade'))")
        # all layers in three launches (select over the fp32 accumulators, fused K/V gather + shadow, accumulator rows),
        # like reference mode's two — not a Python loop of per-layer launches
        n_layers = len(past_key_values)
        base, scaling = _rope_of(past_key_values)
        Ks = [_rows(kv[0]) for kv in past_key_values]
        Vs = [_rows(kv[1]) for kv in past_key_values]
        Ks, Vs = _common_strides(Ks, Vs)
        B, H, _, d = Ks[0].shape
        accs = [self.ext.layers[layer].acc for layer in range(n_layers)]
        if any(a.shape[1] < seq_len for a in accs):
            raise RuntimeError("cascade importance accumulators do not cover the cache")
        if any(a.shape[0] != H for a in accs):
            # grouped-query cache: the accumulators have one row per QUERY head, the cache one plane per KV head — a key's
            # importance is the sum over its group (as in reference mode, apply_token_pruning), every query head's
            # accumulator row then follows its KV head's row map
            return self._prune_cascade_gqa(past_key_values, Ks, Vs, accs, seq_len, num_coming, lo, hi, new_len, base, scaling)
        if any(a.stride(0) != accs[0].stride(0) for a in accs):
            width = max(a.shape[1] for a in accs)
            accs = [torch.nn.functional.pad(a, (0, width - a.shape[1])) for a in accs]
        self.importance_score = [a[:, :seq_len] for a in accs]
        cap = kv_slab.round_capacity(new_len + max(int(num_coming), 0))
        rope = kv_slab.rope_tables(cap, d, Ks[0].dtype, Ks[0].device, base, scaling)
        new_acc = torch.zeros(n_layers, H, max(cap, accs[0].shape[1]), dtype=torch.float32, device=Ks[0].device)
        ranked = self._global_scores([a[:, :seq_len] for a in accs]) if self.token_scope == "global" else accs
        Kn, Vn, Krn, idx = ops.prune_layers(ranked, Ks, Vs, seq_len, lo, hi, self.important_size, capacity=cap, rope=rope,
                                            acc=(accs, [new_acc[layer] for layer in range(n_layers)]))
        out = []
        for layer, (k, v, kr) in enumerate(zip(Kn, Vn, Krn)):
            st = self.ext.layers[layer]
            st.acc = new_acc[layer]
            st.pending_len = 0
            kv_slab.attach(k, v, kr, new_len, base, scaling)
            out.append([k, v])
        self.keep_indices = idx
        self.n_pruned_last = seq_len - new_len
        self.n_pruned_total += self.n_pruned_last
        return out


    def _prune_cascade_gqa(self, past_key_values, Ks, Vs, accs, seq_len, num_coming, lo, hi, new_len, base, scaling):
        n_layers = len(past_key_values)
        B, Hkv, _, d = Ks[0].shape
        group = accs[0].shape[0] // Hkv
        scores = _common_rows([_group_rows(self._live_rows(layer, a[:, :seq_len]), Hkv) for layer, a in enumerate(accs)])
        self.importance_score = scores
        if self.token_scope == "global":
            scores = self._global_scores(scores)
        cap = kv_slab.round_capacity(new_len + max(int(num_coming), 0))
        rope = kv_slab.rope_tables(cap, d, Ks[0].dtype, Ks[0].device, base, scaling)
        Kn, Vn, Krn, idx = ops.prune_layers(scores, Ks, Vs, seq_len, lo, hi, self.important_size, capacity=cap, rope=rope)
        out = []
        for layer, (k, v, kr) in enumerate(zip(Kn, Vn, Krn)):
            st = self.ext.layers[layer]
            rows = idx[layer].repeat_interleave(group, dim=0).contiguous()          # query head h -> KV head h // group
            st.acc = ops.importance_compact(accs[layer], rows, self.start_size, hi, seq_len, max(cap, accs[layer].shape[1]))
            st.pending_len = 0
            kv_slab.attach(k, v, kr, new_len, base, scaling)
            out.append([k, v])
        self.keep_indices = idx
        self.n_pruned_last = seq_len - new_len
        self.n_pruned_total += self.n_pruned_last
        return out

    def _prune_layer_cascade(self, past_key_values, num_coming, attn_score_all):
        """Layer-to-layer cascade (extension, parity unpinned; oracle: layer_cascade_prune): layer l keeps layer_keep[l]
        window tokens, chosen among the tokens layer l-1 just kept; layers end up with different cache lengths.  All layers
        in three launches (ops.prune_layer_cascade): the chain "layer l chooses among what layer l-1 kept" runs per head
        inside one kernel, then one ragged K/V gather and one ragged accumulator gather."""
        ext = self.ext
        n_layers = len(past_key_values)
        start = self.start_size
        Ks, Vs, lens, his, keeps, scores, caps = [], [], [], [], [], [], []
        for layer, (K, V) in enumerate(past_key_values):
            K, V = _rows(K), _rows(V)
            if V.stride() != K.stride():
                K, V = K.contiguous(), V.contiguous()
            L = K.shape[2]
            k_l = ext.layer_keep[layer]
            hi = min(L - self.recent_size + num_coming, L)
            if hi - start < k_l:
                raise ValueError(f"layer {layer}: top-k window [{start},{hi}) holds fewer than {k_l} candidates")
            if self.importance_mode == "cascade":
                score = ext.layers[layer].acc[:, :L]
            else:
                score = ops.importance(attn_score_all[layer])[:, :L]
            if score.shape[0] != K.shape[1]:
                score = self._live_rows(layer, score)
                if self.importance_mode == "cascade":
                    raise NotImplementedError("layer_keep with cascade importance on a grouped-query cache: the accumulators "
                                              "have one row per query head")
                score = _group_rows(score, K.shape[1])        # grouped-query cache: a key's importance = its group's sum
            if self.token_scope == "global":
                # (every layer has its own length here: one reduction per layer; the kernel wants real rows)
                score = self._global_scores([score])[0].contiguous()
            if score.stride(1) != 1:
                score = score.contiguous()
            Ks.append(K); Vs.append(V); lens.append(L); his.append(hi); keeps.append(k_l); scores.append(score)
            caps.append(kv_slab.round_capacity(start + k_l + (L - hi) + max(int(num_coming), 0)))
        self.importance_score = scores
        B, H, _, d = Ks[0].shape
        known = [ext.tok_ids[layer] for layer in range(n_layers)]
        n_known0 = 0 if known[0] is None else known[0].shape[1]
        id_base, appended = ext.next_token_id, lens[0] - n_known0      # the same tokens were appended to every layer
        base, scaling = _rope_of(past_key_values[:1])
        rope = kv_slab.rope_tables(max(caps), d, Ks[0].dtype, Ks[0].device, base, scaling)
        accs = [ext.layers[layer].acc for layer in range(n_layers)] if self.importance_mode == "cascade" else None
        Kn, Vn, Krn, idxs, new_ids, new_accs = ops.prune_layer_cascade(scores, known, id_base, Ks, Vs, lens, his, keeps, start,
                                                                       caps, rope, accs)
        out, pruned = [], 0
        for layer in range(n_layers):
            new_len = Kn[layer].shape[2]
            st = ext.layers[layer]
            if new_accs is not None:
                st.acc = new_accs[layer]
            st.pending_len = 0
            kv_slab.attach(Kn[layer], Vn[layer], Krn[layer], new_len, base, scaling)
            out.append([Kn[layer], Vn[layer]])
            pruned += lens[layer] - new_len
        ext._append_base, ext._append_count = id_base, appended
        ext.after_layer_cascade(new_ids)
        self.keep_indices = idxs                      # a list: the layers keep different numbers of tokens
        self.n_pruned_last = pruned // max(len(out), 1)
        self.n_pruned_total += self.n_pruned_last
        return out


def _rope_of(past_key_values):
    """(base, scaling) the incoming cache was rotated with: the prune rebuilds the rotated shadow at the NEW slot
    positions and must use the tables of the model's rotary embedding (the reference always goes through
    ``self.rotary_emb``, modify_llama.py:89), not a default base."""
    slab = kv_slab.slab_of(past_key_values[0][0])
    if slab is None:
        return 10000.0, None        # foreign tensors: the next forward re-rotates them with the module's tables anyway
    return slab.base, slab.scaling


def _group_rows(score: torch.Tensor, kv_heads: int) -> torch.Tensor:
    H, L = score.shape
    if H % kv_heads:
        raise ValueError(f"{H} score rows cannot be grouped onto {kv_heads} KV heads")
    return score.reshape(kv_heads, H // kv_heads, L).sum(1)


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t if (t.stride(3) == 1 and t.stride(2) == t.shape[3]) else t.contiguous()


def _common_strides(Ks: Sequence[torch.Tensor], Vs: Sequence[torch.Tensor]):
    ref = Ks[0].stride()
    if all(t.stride() == ref for t in list(Ks) + list(Vs)):
        return list(Ks), list(Vs)
    return [t.contiguous() for t in Ks], [t.contiguous() for t in Vs]


def _common_rows(scores: Sequence[torch.Tensor]):
    ok = all(s.stride(1) == 1 and s.stride(0) == scores[0].stride(0) for s in scores)
    return list(scores) if ok else [s.contiguous() for s in scores]
