"""spatten_amd — MI355X (gfx950) implementation of the SpAtten cascade-pruned attention hot path.

Drop-in surface (same names as mit-han-lab/spatten's ``spatten_llm``):
    enable_spatten_llm, SpAttenKVCache, enable_llama_pos_shift_attention,
    llama_pos_shift_attention_forward, apply_rotary_pos_emb_single
backed by hand-written HIP kernels behind the C ABI of ``include/spatten.h``.
"""
from .kv_cache_token_pruning import SpAttenKVCache  # noqa: F401
from .enable_spatten_llm import enable_spatten_llm  # noqa: F401

__all__ = ["SpAttenKVCache", "enable_spatten_llm"]
__version__ = "0.1.0"
