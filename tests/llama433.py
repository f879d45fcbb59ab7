"""A restatement of transformers==4.33.0's Llama decoder stack (the release the reference pins, README.md:46) — test
infrastructure, written from that release's documented behaviour, NOT copied: the installed transformers (5.x) calls its
attention modules with other keyword arguments (position_embeddings, cache objects), so the reference's patched forward
(modify_llama.py:31-40: hidden_states, attention_mask, position_ids, past_key_value, output_attentions, use_cache) does not
run under its decoder layer.  What the 4.33 model does around the attention module, restated:

    LlamaRMSNorm        x * rsqrt(mean(x^2) + eps) in fp32, cast back, * weight
    LlamaMLP            down(silu(gate(x)) * up(x))
    LlamaDecoderLayer   h = x + attn(input_norm(x));  out = h + mlp(post_norm(h))
    LlamaModel.forward  position_ids = arange(P, P + q)[None] with P = past_key_values[0][0].shape[2];
                        attention_mask = additive causal [B, 1, q, N]: 0 for j <= P + i, finfo(dtype).min above;
                        past_key_values = tuple over layers of (K, V); use_cache
    LlamaForCausalLM    lm_head(norm(h)) -> (logits, past_key_values)

The attention class is named ``LlamaAttention`` (the plugin recognises HF's module by class / name, modify_llama.py:171-181)
and carries the 4.33 attribute surface: config.pretraining_tp, hidden_size, num_heads, head_dim, num_key_value_heads,
num_key_value_groups, max_position_embeddings, q/k/v/o_proj, rotary_emb (base 10000, cos/sin cached as [1,1,S,d])."""
from types import SimpleNamespace

import torch
from torch import nn


class LlamaRotaryEmbedding(nn.Module):
    def __init__(self, dim, max_position_embeddings=2048, base=10000.0):
        super().__init__()
        self.dim, self.max_position_embeddings, self.base = dim, max_position_embeddings, base
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    def forward(self, x, seq_len=None):
        t = torch.arange(seq_len, device=x.device, dtype=self.inv_freq.dtype)
        freqs = torch.einsum("i,j->ij", t, self.inv_freq.to(x.device))
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos()[None, None].to(x.dtype), emb.sin()[None, None].to(x.dtype)


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden, eps=1e-6):
        super().__init__()
        self.weight, self.eps = nn.Parameter(torch.ones(hidden)), eps

    def forward(self, x):
        dt = x.dtype
        x = x.float()
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps)
        return self.weight * x.to(dt)


class LlamaMLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.gate_proj, self.up_proj = nn.Linear(hidden, inter, bias=False), nn.Linear(hidden, inter, bias=False)
        self.down_proj = nn.Linear(inter, hidden, bias=False)

    def forward(self, x):
        return self.down_proj(nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaAttention(nn.Module):
    """The 4.33 attribute surface; its own forward is never used (enable_spatten_llm rebinds it)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.hidden_size, self.num_heads = config.hidden_size, config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self.rotary_emb = LlamaRotaryEmbedding(self.head_dim, self.max_position_embeddings, config.rope_theta)

    def forward(self, *a, **k):
        raise RuntimeError("the un-patched attention forward is not restated: call enable_spatten_llm(model, ...) first")


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = LlamaAttention(config)
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False):
        h, _, present = self.self_attn(hidden_states=self.input_layernorm(hidden_states), attention_mask=attention_mask,
                                       position_ids=position_ids, past_key_value=past_key_value,
                                       output_attentions=output_attentions, use_cache=use_cache)
        hidden_states = hidden_states + h
        hidden_states = hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))
        return hidden_states, present


class LlamaModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, input_ids, past_key_values=None, use_cache=True):
        B, q = input_ids.shape
        P = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        x = self.embed_tokens(input_ids)
        position_ids = torch.arange(P, P + q, dtype=torch.long, device=input_ids.device)[None]
        N = P + q
        mask = torch.zeros(B, 1, q, N, dtype=x.dtype, device=x.device)
        if q > 1:
            above = torch.arange(N, device=x.device)[None, :] > (P + torch.arange(q, device=x.device))[:, None]
            mask.masked_fill_(above[None, None], torch.finfo(x.dtype).min)
        presents = []
        for i, layer in enumerate(self.layers):
            x, present = layer(x, attention_mask=mask, position_ids=position_ids,
                               past_key_value=None if past_key_values is None else past_key_values[i], use_cache=use_cache)
            presents.append(present)
        return self.norm(x), tuple(presents)


class LlamaForCausalLM(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    @torch.no_grad()
    def forward(self, input_ids=None, past_key_values=None, use_cache=True):
        h, presents = self.model(input_ids, past_key_values, use_cache)
        return SimpleNamespace(logits=self.lm_head(h), past_key_values=presents)


def tiny_config(hidden=256, heads=4, kv_heads=4, layers=2, inter=512, vocab=131, max_pos=512):
    return SimpleNamespace(model_type="llama", hidden_size=hidden, num_attention_heads=heads, num_key_value_heads=kv_heads,
                           num_hidden_layers=layers, intermediate_size=inter, vocab_size=vocab, rms_norm_eps=1e-6,
                           max_position_embeddings=max_pos, rope_theta=10000.0, pretraining_tp=1, rope_scaling=None)
