"""torch-CPU mirror of the reference's hot path, op for op (TEST INFRASTRUCTURE / CPU BASELINE ONLY — see the header of
spatten_oracle.py for who may import oracle/).

The numpy oracle restates the arithmetic; this file restates the COST: the same sequence of eager torch ops the
reference's Python issues per decode step and per prune event, so that timing it on the GPU box's host cores gives the
"reference CPU path" figure (the reference's own source cannot travel).  ``tools/cpu_port_check.py`` times it against
the imported reference in the build container (same machine, same thread counts) and records the ratio under
profiles/ — the ±20 % acceptance check of SURVEY §8d.

Reference lines mirrored: spatten_llm/pos_shift/modify_llama.py:21-28 (rotation), :86-147 (attention core between
the projections and o_proj), spatten_llm/kv_cache_token_pruning.py:42-96 (prune event).
"""
import math

import torch


def rotary_table(n, d, dtype, base=10000.0):
    """transformers 4.33 LlamaRotaryEmbedding: [1,1,n,d] cos / sin in the model dtype (modify_llama.py:89)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    f = torch.einsum("i,j->ij", torch.arange(n, dtype=torch.float32), inv_freq)
    e = torch.cat((f, f), dim=-1)
    return e.cos()[None, None].to(dtype), e.sin()[None, None].to(dtype)


def rotate(x, cos, sin, pos):
    """modify_llama.py:21-28: table rows gathered by position, x*cos + rotate_half(x)*sin."""
    c = cos.squeeze(1).squeeze(0)[pos].unsqueeze(1)
    s = sin.squeeze(1).squeeze(0)[pos].unsqueeze(1)
    h = x.shape[-1] // 2
    return (x * c) + (torch.cat((-x[..., h:], x[..., :h]), dim=-1) * s)


def decode_core(q, k_new, v_new, past_k, past_v, cos, sin):
    """One decode step of modify_llama.py:86-147 (q_len = 1, MHA): q / k_new / v_new [B,H,1,d], past [B,H,P,d]
    un-rotated.  Returns (attn_output [B,1,H*d], stash [B,H,1,N], (K, V))."""
    B, H, _, d = q.shape
    N = past_k.shape[2] + 1
    q = rotate(q, cos, sin, torch.full((B, 1), N - 1, dtype=torch.long))                       # :92
    k = torch.cat([past_k, k_new], dim=2)                                                      # :95-98
    v = torch.cat([past_v, v_new], dim=2)
    kr = rotate(k, cos, sin, torch.arange(N)[None])                                            # :103-104
    w = torch.matmul(q, kr.transpose(2, 3)) / math.sqrt(d)                                     # :111-113
    stash = w.detach().clone()                                                                 # :116-119
    w = w + torch.zeros(B, 1, 1, N, dtype=q.dtype)                                             # :132 (HF mask at q_len = 1)
    w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)                              # :135-137
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, 1, H * d)                   # :138-147
    return o, stash, (k, v)


def prune_layer(K, V, stash, start, recent, important, num_coming):
    """One layer of kv_cache_token_pruning.py:51-96 (B = 1): importance, window top-k, sorted indices, bool mask via
    the host, boolean gather, three-way concat."""
    L = K.size(2)
    score = stash.sum(0).sum(1)                                                                # :51
    sel = score[:, start:L - recent + num_coming]                                              # :59
    _, idx = torch.topk(sel, important, dim=-1)                                                # :60-61
    idx = idx.sort().values + start                                                            # :62-63
    m = torch.zeros(score.shape, dtype=torch.bool).scatter(-1, idx, 1).cpu()                   # :64-66
    Ki = K.squeeze()[m].view(1, K.size(1), -1, K.size(-1))                                     # :68-69
    Vi = V.squeeze()[m].view(1, V.size(1), -1, V.size(-1))
    tail = L - recent + num_coming
    return (torch.cat([K[:, :, :start], Ki, K[:, :, tail:L]], dim=2),                           # :72-96
            torch.cat([V[:, :, :start], Vi, V[:, :, tail:L]], dim=2))
