"""The plugin under a REAL decoder stack: a 2-layer random-weight LlamaForCausalLM — RMSNorm, SwiGLU MLP,
residuals, lm_head, the mask / position_ids / past_key_values exactly as transformers 4.33's LlamaModel.forward builds them
(tests/llama433.py: a restatement, because the installed transformers 5.x decoder layer passes other arguments than the
reference's patched forward takes, modify_llama.py:31-40) — through enable_spatten_llm and the reference's caller protocol
(run_spatten_llama.py:18-87: prefill, greedy decode, prune at the turn boundary from the last step's stashes, next prompt).
GPU (HIP kernels) vs a numpy replica of the same stack built on the oracle's restatement of the reference attention."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests import llama433

pytestmark = pytest.mark.gpu
START, IMPORTANT, RECENT, MAX_GEN = 4, 24, 24, 8


class Replica:
    def __init__(self, model):
        g = lambda t: t.detach().float().cpu().numpy()
        m = model.model
        self.cfg = model.config
        self.emb, self.lm, self.norm = g(m.embed_tokens.weight), g(model.lm_head.weight), g(m.norm.weight)
        self.layers = []
        for l in m.layers:
            a = l.self_attn
            self.layers.append(dict(q=g(a.q_proj.weight), k=g(a.k_proj.weight), v=g(a.v_proj.weight), o=g(a.o_proj.weight),
                                    gate=g(l.mlp.gate_proj.weight), up=g(l.mlp.up_proj.weight), down=g(l.mlp.down_proj.weight),
                                    n1=g(l.input_layernorm.weight), n2=g(l.post_attention_layernorm.weight)))
        self.stash = [None] * len(self.layers)

    def rms(self, x, w):
        x = x.astype(np.float32)
        return w * (x / np.sqrt((x * x).mean(-1, keepdims=True) + np.float32(self.cfg.rms_norm_eps)))

    def forward(self, ids, past):
        c = self.cfg
        H, Hkv = c.num_attention_heads, c.num_key_value_heads
        D = c.hidden_size // H
        B, q = ids.shape
        P = 0 if past is None else past[0][0].shape[2]
        N = P + q
        pos = np.tile(np.arange(P, N)[None], (B, 1))
        mask = orc.causal_mask(B, q, N, "f32")
        x = self.emb[ids]
        new_past = []
        for i, w in enumerate(self.layers):
            hn = self.rms(x, w["n1"])
            sp = lambda t, h: np.swapaxes(t.reshape(B, q, h, D), 1, 2)
            o, stash, kv = orc.attention_core(sp(hn @ w["q"].T, H), sp(hn @ w["k"].T, Hkv), sp(hn @ w["v"].T, Hkv),
                                              None if past is None else past[i][0], None if past is None else past[i][1],
                                              pos, mask, "f32")
            self.stash[i] = stash
            x = x + o @ w["o"].T
            hn = self.rms(x, w["n2"])
            gate = hn @ w["gate"].T
            x = x + ((gate / (1 + np.exp(-gate))) * (hn @ w["up"].T)) @ w["down"].T
            new_past.append(kv)
        return self.rms(x, self.norm) @ self.lm.T, new_past


def _greedy_gpu(model, ids, past):
    out = model(input_ids=ids, past_key_values=past, use_cache=True)
    toks = [int(out.logits[:, -1].argmax(-1)[0])]
    for _ in range(MAX_GEN - 1):
        out = model(input_ids=torch.tensor([[toks[-1]]], device="cuda"), past_key_values=out.past_key_values, use_cache=True)
        toks.append(int(out.logits[:, -1].argmax(-1)[0]))
    return toks, out.past_key_values, out.logits


def _greedy_ref(ref, ids, past):
    logits, past = ref.forward(ids, past)
    toks = [int(logits[0, -1].argmax())]
    for _ in range(MAX_GEN - 1):
        logits, past = ref.forward(np.asarray([[toks[-1]]]), past)
        toks.append(int(logits[0, -1].argmax()))
    return toks, past, logits


@pytest.mark.parametrize("kv_heads,auto_graph", [(4, False), (4, True)])
def test_plugin_under_a_llama433_decoder_stack_matches_the_numpy_replica(kv_heads, auto_graph, capsys):
    import transformers
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(0)
    cfg = llama433.tiny_config(kv_heads=kv_heads)
    model = llama433.LlamaForCausalLM(cfg).cuda().float()
    ref = Replica(model)
    cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, auto_graph=auto_graph)       # run_spatten_llama.py:110-115
    mods = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]
    assert len(mods) == cfg.num_hidden_layers
    rng = np.random.default_rng(5)
    prompts = [rng.integers(0, cfg.vocab_size, size=n)[None] for n in (44, 21, 30)]
    past_g = past_r = None
    pruned = 0
    for turn, prompt in enumerate(prompts):
        if turn > 0:                                                                   # :71-83
            space = prompt.shape[1] + MAX_GEN
            n0 = past_g[0][0].shape[2]
            past_g = cache.apply_token_pruning(past_g, space, [m.attn_scores for m in mods])
            pruned += n0 - past_g[0][0].shape[2]
            past_r, idxs = orc.apply_token_pruning(past_r, space, ref.stash, START, RECENT, IMPORTANT, "f32")
            if idxs is not None:
                assert np.array_equal(cache.keep_indices.cpu().numpy(), np.stack(idxs)), f"turn {turn}: kept indices"
        tg, past_g, lg = _greedy_gpu(model, torch.from_numpy(prompt).cuda(), past_g)
        tr, past_r, lr = _greedy_ref(ref, prompt, past_r)
        assert tg == tr, f"turn {turn}: tokens {tg} vs {tr}"
        np.testing.assert_allclose(lg.float().cpu().numpy(), lr, atol=3e-4, rtol=3e-4)
        assert past_g[0][0].shape[2] == past_r[0][0].shape[2]
        for (kg, vg), (kr, vr) in zip(past_g, past_r):
            np.testing.assert_allclose(kg.cpu().numpy(), kr, atol=2e-5, rtol=2e-5)
            np.testing.assert_allclose(vg.cpu().numpy(), vr, atol=2e-5, rtol=2e-5)
    assert pruned > 0
    print(f"[hf-decoder] restated transformers 4.33 stack (installed: {transformers.__version__}); auto_graph={auto_graph}: "
          f"3 turns, {pruned} tokens pruned, tokens and logits equal to the numpy replica")


def test_plugin_under_the_decoder_stack_in_bf16_keeps_the_reference_tokens():
    """The same stack in bf16 (the deployment dtype): the HIP path against the fp32 replica — greedy tokens of the first turn
    and logits within the 16-bit tolerance."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(1)
    cfg = llama433.tiny_config()
    model = llama433.LlamaForCausalLM(cfg).cuda().float()
    ref = Replica(model)
    model = model.to(torch.bfloat16)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        enable_spatten_llm(model, START, IMPORTANT, RECENT)
    prompt = np.random.default_rng(6).integers(0, cfg.vocab_size, size=40)[None]
    out = model(input_ids=torch.from_numpy(prompt).cuda(), past_key_values=None, use_cache=True)
    lr, _ = ref.forward(prompt, None)
    np.testing.assert_allclose(out.logits.float().cpu().numpy(), lr, atol=0.15, rtol=0.1)
