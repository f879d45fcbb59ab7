"""Plugin surface without a GPU: the patch traversal over a real transformers LlamaForCausalLM, the rotary-table
parameters taken from the module, the extension arguments, the harness loaders."""
import os
import types

import pytest
import torch

from spatten_amd import enable_spatten_llm
from spatten_amd.pos_shift import modify_llama as ml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hf_llama():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=3,
                                   num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=64)
    return transformers.LlamaForCausalLM(cfg)


def test_patch_traversal_on_a_real_hf_llama(hf_llama, capsys):
    """modify_llama.py:171-181 walks ``model._modules`` depth-first and rebinds ``forward`` on every LlamaAttention
    INSTANCE: exercised on the installed transformers' own module tree (the patched forward targets the 4.33 decoder
    layer's calling convention; here only the traversal is under test)."""
    from transformers.models.llama.modeling_llama import LlamaAttention
    attn = [m for m in hf_llama.modules() if isinstance(m, LlamaAttention)]
    assert len(attn) == 3
    others = [m for m in hf_llama.modules() if not isinstance(m, LlamaAttention)]
    before = {id(m): m.forward for m in others}
    cache = enable_spatten_llm(hf_llama, start_size=4, important_size=8, recent_size=8)
    for m in attn:
        assert isinstance(m.forward, types.MethodType) and m.forward.__func__ is ml.llama_pos_shift_attention_forward
        assert m.forward.__self__ is m
    for m in others:                                   # nothing else was touched
        assert m.forward == before[id(m)]
    assert ml.attention_modules(hf_llama) == attn      # model.modules() order = layer order (run_spatten_llama.py:74-77)
    assert (cache.start_size, cache.important_size, cache.recent_size, cache.cache_size) == (4, 8, 8, 20)
    assert cache.k_seq_dim == cache.v_seq_dim == 2
    assert "SpAttenKVCache: keep start: 4" in capsys.readouterr().out
    # extension modes register their per-layer state on the same modules
    cache2 = enable_spatten_llm(hf_llama, 4, 8, 8, importance_mode="cascade", head_keep=[4, 3, 3], pq_threshold=0.05)
    assert [m._spatten_ext[1] for m in attn] == [0, 1, 2] and all(m._spatten_ext[0] is cache2.ext for m in attn)
    assert cache2.ext.head_keep == [4, 3, 3] and cache2.ext.cascade and cache2.ext.pq_threshold == 0.05


def test_non_llama_models_are_rejected_like_the_reference():
    m = types.SimpleNamespace(config=types.SimpleNamespace(model_type="gpt2"))
    with pytest.raises(ValueError, match="got gpt2"):              # enable_spatten_llm.py:13-14
        enable_spatten_llm(m, 4, 8, 8)


def test_extension_arguments_are_validated(hf_llama):
    with pytest.raises(ValueError, match="one entry per layer"):
        enable_spatten_llm(hf_llama, 4, 8, 8, head_keep=[4, 3])
    with pytest.raises(ValueError, match="must not grow"):
        enable_spatten_llm(hf_llama, 4, 8, 8, head_keep=[2, 3, 3])
    with pytest.raises(ValueError, match="cannot be combined"):
        enable_spatten_llm(hf_llama, 4, 8, 8, pq_threshold=0.1, local_v_keep=0.5)
    with pytest.raises(ValueError, match="fraction"):
        enable_spatten_llm(hf_llama, 4, 8, 8, local_v_keep=1.5)
    with pytest.raises(ValueError, match="auto_graph"):                 # the one mode whose decode step is not capturable
        enable_spatten_llm(hf_llama, 4, 8, 8, local_v_keep=0.5, importance_mode="cascade", auto_graph=True)
    assert not hasattr(hf_llama, "_spatten_auto_graph")
    with pytest.raises(ValueError, match="pq_threshold"):               # round 4: bit profiles, fused step
        enable_spatten_llm(hf_llama, 4, 8, 8, pq_profile=(4, 8))
    with pytest.raises(ValueError, match="one of"):
        enable_spatten_llm(hf_llama, 4, 8, 8, pq_threshold=0.1, pq_profile=(5, 8))
    with pytest.raises(ValueError, match="cascade"):
        enable_spatten_llm(hf_llama, 4, 8, 8, pq_threshold=0.1, pq_profile=(4, 8), importance_mode="cascade")
    with pytest.raises(ValueError, match="fused_step"):
        enable_spatten_llm(hf_llama, 4, 8, 8, fused_step=True)
    c = enable_spatten_llm(hf_llama, 4, 8, 8, importance_mode="cascade", prefill_stash=False)     # stash-free cascade
    assert c.ext.cascade and c.ext.prefill_wants_lse(torch.bfloat16, 128, 64, False)
    assert not c.ext.prefill_wants_lse(torch.float32, 128, 64, False) and not c.ext.prefill_wants_lse(torch.bfloat16, 128, 64, True)


def test_rope_parameters_come_from_the_module():
    ns = types.SimpleNamespace
    plain = ns(rotary_emb=ns(base=10000.0), config=ns())
    assert ml._rope_params(plain) == (10000.0, None)
    code = ns(rotary_emb=None, config=ns(rope_theta=1e6, rope_scaling=None))          # CodeLlama: rope_theta 1e6
    assert ml._rope_params(code) == (1e6, None)
    lin = ns(rotary_emb=ns(base=10000.0), config=ns(rope_scaling={"type": "linear", "factor": 4.0}))
    assert ml._rope_params(lin) == (10000.0, ("linear", 4.0))

    class LlamaLinearScalingRotaryEmbedding:        # 4.33 class name, scaling_factor attribute
        base, scaling_factor = 10000.0, 2.0
    assert ml._rope_params(ns(rotary_emb=LlamaLinearScalingRotaryEmbedding(), config=ns())) == (10000.0, ("linear", 2.0))

    class LlamaDynamicNTKScalingRotaryEmbedding:
        base, scaling_factor = 10000.0, 2.0
    with pytest.raises(NotImplementedError):
        ml._rope_params(ns(rotary_emb=LlamaDynamicNTKScalingRotaryEmbedding(), config=ns()))
    with pytest.raises(NotImplementedError):
        ml._rope_params(ns(rotary_emb=None, config=ns(rope_scaling={"type": "dynamic", "factor": 2.0})))


def test_linear_scaled_table_is_the_4_33_recipe():
    from spatten_amd import ops
    cos, sin = ops.rope_table(16, 8, torch.float32, "cpu", base=500.0, scaling=("linear", 4.0))
    inv = 1.0 / (500.0 ** (torch.arange(0, 8, 2).float() / 8))
    t = torch.arange(16, dtype=torch.float32) / 4.0                 # LlamaLinearScalingRotaryEmbedding: t / scaling_factor
    want = torch.einsum("i,j->ij", t, inv)
    assert torch.equal(cos, want.cos()) and torch.equal(sin, want.sin())


def test_mt_bench_loader_flattens_turns():
    from spatten_llm.utils import load_jsonl, load_mt_bench_prompts      # the reference's import path
    path = os.path.join(ROOT, "tests", "golden", "mt_bench_sample.jsonl")
    rows = load_jsonl(path)
    assert [r["question_id"] for r in rows] == [81, 82, 83]
    prompts = load_mt_bench_prompts(path)
    assert len(prompts) == 5 and prompts[1].startswith("Now shorten") and prompts[-1] == "What is 7 times 6?"


def test_bounded_caches():
    from spatten_amd.ops import _LRU
    c = _LRU(3)
    for i in range(5):
        c.put(i, i * i)
    assert c.get(0) is None and c.get(1) is None and c.get(2) == 4
    c.put(9, 81)                                    # 3 is now the oldest (2 was just touched)
    assert c.get(3) is None and sorted(c.values()) == [4, 16, 81]


def test_fuse_qkv_projections_stacks_weights_as_views():
    import torch
    from torch import nn
    from types import SimpleNamespace
    from spatten_amd.pos_shift.modify_llama import fuse_qkv_projections

    class M(nn.Module):
        def __init__(self, tp=1, bias=True):
            super().__init__()
            self.config = SimpleNamespace(pretraining_tp=tp)
            self.q_proj, self.k_proj, self.v_proj = nn.Linear(16, 16, bias=bias), nn.Linear(16, 8, bias=bias), nn.Linear(16, 8, bias=bias)

    m = M()
    x = torch.randn(3, 16)
    want = [m.q_proj(x), m.k_proj(x), m.v_proj(x)]
    assert fuse_qkv_projections(m)
    w, b, nq, nk = m._spatten_qkv
    assert w.shape == (32, 16) and b.shape == (32,) and (nq, nk) == (16, 8)
    assert m.k_proj.weight.data_ptr() == w[16:].data_ptr()                 # a view of the stacked weight
    qkv = torch.nn.functional.linear(x, w, b)
    for got, ref in zip((qkv[:, :16], qkv[:, 16:24], qkv[:, 24:]), want):
        torch.testing.assert_close(got, ref)
    torch.testing.assert_close(m.v_proj(x), want[2])                       # the modules themselves still work
    assert not fuse_qkv_projections(M(tp=2))                               # pretraining_tp > 1: left alone
    mixed = M()
    mixed.k_proj = nn.Linear(16, 8, bias=False)
    assert not fuse_qkv_projections(mixed)


def test_auto_graph_leaves_every_other_call_to_the_original_forward():
    """spatten_amd.graph.auto_graph takes over only single-token keyword calls on device tensors under no_grad; on this
    (GPU-less) box every call must reach the original forward untouched — prefill, positional calls, calls with a mask."""
    import torch
    from spatten_amd.graph import auto_graph

    calls = []

    class M:
        def forward(self, *a, **kw):
            calls.append((a, sorted(kw)))
            return "orig"

    m = auto_graph(M())
    ids1, ids5 = torch.zeros(1, 1, dtype=torch.long), torch.zeros(1, 5, dtype=torch.long)
    past = [[torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4)]]
    with torch.no_grad():
        assert m.forward(input_ids=ids5, past_key_values=None, use_cache=True) == "orig"          # prefill
        assert m.forward(input_ids=ids1, past_key_values=past, use_cache=True) == "orig"          # CPU tensors: not taken
        assert m.forward(ids1, past) == "orig"                                                    # positional
        assert m.forward(input_ids=ids1, past_key_values=past, use_cache=True, attention_mask=torch.ones(1, 4)) == "orig"
    assert len(calls) == 4 and m._spatten_auto_graph["graph"] is None
