"""Plugin entry point — mirror of spatten_llm/enable_spatten_llm.py:5-23."""
from .kv_cache_token_pruning import SpAttenKVCache

__all__ = ["enable_spatten_llm"]


def enable_spatten_llm(model, start_size, important_size, recent_size):
    if "llama" in model.config.model_type:                                   # :6
        k_seq_dim = v_seq_dim = 2
        from .pos_shift.modify_llama import enable_llama_pos_shift_attention

        enable_llama_pos_shift_attention(model)
    else:
        raise ValueError(f"got {model.config.model_type}")                    # :13-14
    return SpAttenKVCache(start_size=start_size, important_size=important_size, recent_size=recent_size,
                          k_seq_dim=k_seq_dim, v_seq_dim=v_seq_dim)
