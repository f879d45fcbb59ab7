"""Plugin entry point of the drop-in surface.

Contract (reference: spatten_llm/enable_spatten_llm.py:5-23): ``enable_spatten_llm(model, start_size,
important_size, recent_size)`` patches the model's attention modules for cache-relative RoPE + score stashing and
returns the ``SpAttenKVCache`` that prunes its KV cache; model families the path does not cover raise
``ValueError("got <model_type>")``.
"""
from typing import Callable, Dict, Tuple

from .kv_cache_token_pruning import SpAttenKVCache

__all__ = ["enable_spatten_llm"]


def _patch_llama(model) -> Tuple[int, int]:
    from .pos_shift.modify_llama import enable_llama_pos_shift_attention

    enable_llama_pos_shift_attention(model)
    return 2, 2            # K and V are [B, H, L, d]: the sequence axis is dim 2


# model_type substring -> (patcher returning the (k_seq_dim, v_seq_dim) of that family's cache layout)
_FAMILIES: Dict[str, Callable] = {"llama": _patch_llama}


def enable_spatten_llm(model, start_size, important_size, recent_size, importance_mode="reference",
                       prefill_stash=True, assume_causal=False):
    """``importance_mode="cascade"`` (extension, parity unpinned) makes the patched forward accumulate softmax
    probabilities per (layer, head, key) and the returned cache prune by them instead of by the last step's logits.

    ``prefill_stash=False`` (extension): forwards with ``q_len > 1`` do not materialise ``self.attn_scores``
    ([B,H,q,N]: 4 GiB per layer at q = N = 8192; the reference writes it on every forward, modify_llama.py:116-119) —
    it is set to ``None`` there.  The caller protocol (run_spatten_llama.py:71-79) prunes from the LAST DECODE step's
    stash, which is still written.  ``assume_causal=True``: a non-None HF mask at ``q_len > 1`` is taken to be the
    causal mask and not read (tiles above the diagonal are skipped)."""
    model_type = model.config.model_type
    patch = next((fn for key, fn in _FAMILIES.items() if key in model_type), None)
    if patch is None:
        raise ValueError(f"got {model_type}")
    k_dim, v_dim = patch(model)
    cache = SpAttenKVCache(start_size=start_size, recent_size=recent_size, important_size=important_size,
                           k_seq_dim=k_dim, v_seq_dim=v_dim, importance_mode=importance_mode)
    if not prefill_stash or assume_causal:
        from .pos_shift.modify_llama import attention_modules

        if importance_mode == "cascade" and not prefill_stash:
            raise ValueError("importance_mode='cascade' accumulates from the prefill stash: prefill_stash must stay True")
        for m in attention_modules(model):
            m.spatten_prefill_stash = bool(prefill_stash)
            m.spatten_assume_causal = bool(assume_causal)
    if importance_mode == "cascade":
        from .pos_shift.modify_llama import attention_modules

        mods = attention_modules(model)                            # model.modules() order = layer order (:74-77)
        cache._cascade_modules = mods
        for layer, m in enumerate(mods):
            m._spatten_cascade = (cache, layer)
    return cache
