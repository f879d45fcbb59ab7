from spatten_amd.pos_shift.modify_llama import (  # noqa: F401
    apply_rotary_pos_emb_single,
    enable_llama_pos_shift_attention,
    llama_pos_shift_attention_forward,
)

__all__ = ["enable_llama_pos_shift_attention"]
