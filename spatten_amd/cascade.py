"""SpAtten's cascade semantics on top of the drop-in path — PARITY UNPINNED: the reference's
Python implements none of this (its importance is the last step's raw logits, kv_cache_token_pruning.py:51); the
rules below restate the paper / RTL control flow and are checked against oracle/spatten_oracle.py.

* ``CascadeImportance``  cumulative importance = running sum of softmax probabilities per (head, key)
  (README.md:11; trace flag ``if_accumulate_importance``), carried through prunes with the cache.
* ``local_v_decode``     local V pruning (SpAttenController.scala:546-558, 591-612): per head keep the ``keep``
  largest probabilities, fetch only those V rows, no renormalisation.
* ``HeadPruner``         head pruning (README.md:21): cumulative sum |attn_out_h|, keep the top heads; the decode
  kernel then launches only the kept heads.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops


class CascadeImportance:
    def __init__(self, layers: int, heads: int, capacity: int, device):
        self.acc: List[torch.Tensor] = [torch.zeros(heads, capacity, dtype=torch.float32, device=device) for _ in range(layers)]

    def accumulate(self, layer: int, stash: torch.Tensor, lse: Optional[torch.Tensor] = None,
                   mask: Optional[torch.Tensor] = None, causal: bool = False):
        """stash [B,H,q,L] of this layer's last forward; lse [B,H,q,2] when the kernel already produced it."""
        L = stash.shape[-1]
        if self.acc[layer].shape[1] < L:
            grown = torch.zeros(self.acc[layer].shape[0], 2 * L, dtype=torch.float32, device=stash.device)
            grown[:, :self.acc[layer].shape[1]] = self.acc[layer]
            self.acc[layer] = grown
        ops.importance_accumulate(self.acc[layer], stash, lse, mask, causal)

    def select(self, layer: int, L: int, lo: int, hi: int, k: int) -> torch.Tensor:
        """Kept positions of the window [lo, hi): top-k of the ACCUMULATED importance (fp32), ascending."""
        return ops.topk_select(self.acc[layer][:, :L], lo, hi, k)

    def compact(self, layer: int, idx: torch.Tensor, start: int, tail_lo: int, L: int, capacity: Optional[int] = None):
        self.acc[layer] = ops.importance_compact(self.acc[layer], idx, start, tail_lo, L, capacity or self.acc[layer].shape[1])


def local_v_decode(q, kr_cache, v_cache, kv_len, cos, sin, pos_q, keep, mask=None, workspace=None, out=None, stash=None,
                   lse=None, three_launches: bool = False, layout: int = 0, k_new=None, v_new=None, k_cache=None):
    """Decode step with local V pruning: scores + (max, sum) without touching V, per-(b,h) top-``keep`` of the logits
    (same order as the probabilities), P·V over the kept rows only — ONE launch (ops.attn_decode_local_v, round 4); with an
    additive ``mask``, splits beyond 16384 rows or ``three_launches=True`` the r02 form: three dependent launches.
    Returns (out [B,H*d], stash); ``stash`` [B,H,>=kv_len] / ``lse`` [B,H,2] may be caller buffers.  ``k_new`` / ``v_new``
    [B,Hkv,d]: the step's append (row kv_len - 1 of k_cache / kr_cache / v_cache) — inside the one launch (round 5), by
    ``ops.kv_append`` in front of the three-launch form."""
    B, H, d = q.shape
    if stash is None:
        stash = torch.empty(B, H, kv_len, dtype=q.dtype, device=q.device)
    if lse is None:
        lse = torch.empty(B, H, 2, dtype=torch.float32, device=q.device)
    if mask is None and not three_launches and d in (64, 128):
        try:
            out = ops.attn_decode_local_v(q, kr_cache, v_cache, kv_len, cos, sin, pos_q, min(keep, kv_len), stash, out=out,
                                          lse=lse, layout=layout, k_new=k_new, v_new=v_new, k_cache=k_cache)
            return out, stash
        except NotImplementedError:
            pass
    if k_new is not None:
        ops.kv_append(k_new[:, :, None], v_new[:, :, None], k_cache, kr_cache, v_cache, kv_len - 1, cos, sin)
    ops.attn_decode(q, None, kr_cache, v_cache, kv_len, cos, sin, pos_q, mask=mask, scores=stash, lse=lse,
                    scores_only=True, workspace=workspace)
    keep = min(keep, kv_len)
    logits = stash[:, :, :kv_len] if mask is None else (stash[:, :, :kv_len] + mask[:, None, :])
    idx = ops.topk_select(logits.reshape(B * H, kv_len), 0, kv_len, keep)
    out = ops.pv_gather(stash, lse, v_cache, idx, mask=mask, out=out)
    return out, stash


class HeadPruner:
    def __init__(self, heads: int, device):
        self.heads = heads
        self.scores = torch.zeros(heads, dtype=torch.float32, device=device)

    def observe(self, attn_out: torch.Tensor):
        """attn_out [B,q,H*d] (before o_proj) of one layer-step."""
        ops.head_scores(attn_out, self.heads, self.scores)

    def select(self, keep: int) -> torch.Tensor:
        """int32 ascending ids of the ``keep`` highest-scoring heads (ties: lowest id) — identical on every rank
        once the scores have been all-gathered (parallel.HeadParallel.gather_head_scores)."""
        return ops.topk_select(self.scores[None, :], 0, self.heads, keep)[0].contiguous()
