"""End-to-end parity of the whole plugin under the reference's caller protocol (run_spatten_llama.py:18-87):
enable_spatten_llm -> multi-turn {prune at turn boundary, prefill, greedy decode} on a tiny random Llama-like stack,
GPU (patched forward + SpAttenKVCache on the HIP kernels) vs a numpy replica built from the oracle's restatement of
the reference ops.  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from oracle import spatten_oracle as orc

pytestmark = pytest.mark.gpu

L, H, D, VOCAB = 2, 4, 64, 97
HID = H * D
START, IMPORTANT, RECENT, MAX_GEN = 4, 20, 24, 8


class LlamaAttention(nn.Module):          # duck-typed by class name, like HF's module
    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(pretraining_tp=1)
        self.num_heads = self.num_key_value_heads = H
        self.num_key_value_groups = 1
        self.head_dim, self.hidden_size = D, HID
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(HID, HID, bias=False) for _ in range(4))


class Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = LlamaAttention()


class TinyLlama(nn.Module):
    def __init__(self):
        super().__init__()
        self.config = SimpleNamespace(model_type="llama")
        self.embed = nn.Embedding(VOCAB, HID)
        self.layers = nn.ModuleList([Layer() for _ in range(L)])
        self.lm_head = nn.Linear(HID, VOCAB, bias=False)

    @torch.no_grad()
    def forward(self, ids, past):
        """HF 4.33 LlamaModel protocol: position_ids = arange(P, P+q), additive causal mask [B,1,q,N]."""
        B, q = ids.shape
        P = 0 if past is None else past[0][0].shape[2]
        N = P + q
        pos = torch.arange(P, N, device=ids.device)[None]
        mask = torch.from_numpy(orc.causal_mask(B, q, N, "f32")).to(ids.device)
        x = self.embed(ids)
        new_past = []
        for i, layer in enumerate(self.layers):
            a, _, kv = layer.self_attn(x, attention_mask=mask, position_ids=pos,
                                       past_key_value=None if past is None else past[i], use_cache=True)
            x = x + a
            new_past.append(kv)
        return self.lm_head(x), new_past


class NumpyReplica:
    """Same weights, reference semantics restated by the oracle (cat + re-rotate all keys every step)."""

    def __init__(self, model, cascade=False):
        self.cascade = cascade
        g = lambda t: t.detach().cpu().numpy().astype(np.float32)
        self.emb, self.lm = g(model.embed.weight), g(model.lm_head.weight)
        self.w = [{n: g(getattr(l.self_attn, n).weight) for n in ("q_proj", "k_proj", "v_proj", "o_proj")} for l in model.layers]
        self.stash = [None] * L
        self.acc = None          # cascade mode: [L][H, len] accumulated probabilities

    def accumulate(self, i, stash, mask):
        if self.acc is None:
            self.acc = [np.zeros((H, 0), np.float32) for _ in range(L)]
        n = stash.shape[-1]
        if self.acc[i].shape[1] < n:
            self.acc[i] = np.concatenate([self.acc[i], np.zeros((H, n - self.acc[i].shape[1]), np.float32)], 1)
        m = np.where(mask < 0, -np.inf, 0).astype(np.float32)
        self.acc[i] = orc.cascade_importance_accumulate(self.acc[i], stash, m)

    def forward(self, ids, past):
        B, q = ids.shape
        P = 0 if past is None else past[0][0].shape[2]
        N = P + q
        pos = np.tile(np.arange(P, N)[None], (B, 1))
        mask = orc.causal_mask(B, q, N, "f32")
        x = self.emb[ids]
        new_past = []
        sp = lambda t: np.swapaxes(t.reshape(B, q, H, D), 1, 2)
        for i, w in enumerate(self.w):
            o, stash, kv = orc.attention_core(sp(x @ w["q_proj"].T), sp(x @ w["k_proj"].T), sp(x @ w["v_proj"].T),
                                              None if past is None else past[i][0], None if past is None else past[i][1],
                                              pos, mask, "f32")
            self.stash[i] = stash
            if self.cascade:
                self.accumulate(i, stash, mask)
            x = x + o @ w["o_proj"].T
            new_past.append(kv)
        return x @ self.lm.T, new_past


def greedy(fwd, ids, past, to_ids):
    logits, past = fwd(ids, past)
    toks = [int(np.asarray(logits[:, -1].argmax(-1).cpu() if torch.is_tensor(logits) else logits[:, -1].argmax(-1))[0])]
    for _ in range(MAX_GEN - 1):
        logits, past = fwd(to_ids([[toks[-1]]]), past)
        toks.append(int(np.asarray(logits[:, -1].argmax(-1).cpu() if torch.is_tensor(logits) else logits[:, -1].argmax(-1))[0]))
    return toks, past, logits


def test_multi_turn_protocol_matches_reference_semantics(capsys):
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(0)
    model = TinyLlama().cuda().float()
    for p in model.parameters():
        p.data.mul_(0.6)
    ref = NumpyReplica(model)
    kv_cache = enable_spatten_llm(model, start_size=START, important_size=IMPORTANT, recent_size=RECENT)   # :110-115
    attn_modules = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]

    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, VOCAB, size=n)[None] for n in (40, 25, 30, 12)]
    past_g = past_r = None
    pruned_g = pruned_r = 0
    for turn, prompt in enumerate(prompts):
        if turn > 0:                                                                  # prune event (:71-83)
            space_needed = prompt.shape[1] + MAX_GEN
            scores = [m.attn_scores for m in attn_modules]                            # :74-77
            n_prev = past_g[0][0].size(2)
            past_g = kv_cache.apply_token_pruning(past_g, space_needed, scores)       # :79
            pruned_g += n_prev - past_g[0][0].size(2)
            new_r, idxs = orc.apply_token_pruning(past_r, space_needed, ref.stash, START, RECENT, IMPORTANT, "f32")
            pruned_r += past_r[0][0].shape[2] - new_r[0][0].shape[2]
            if idxs is not None:
                assert np.array_equal(kv_cache.keep_indices.cpu().numpy(), np.stack(idxs)), f"turn {turn}: kept indices"
            past_r = new_r
            for lg, lr in zip(past_g, past_r):     # same rows kept (values come from two different GEMMs: fp32 round-off)
                np.testing.assert_allclose(lg[0].cpu().numpy(), lr[0], atol=1e-5, rtol=1e-5)
                np.testing.assert_allclose(lg[1].cpu().numpy(), lr[1], atol=1e-5, rtol=1e-5)
        tg, past_g, lg_ = greedy(lambda i, p: model(i, p), torch.from_numpy(prompt).cuda(), past_g,
                                 lambda a: torch.tensor(a, device="cuda"))
        tr, past_r, lr_ = greedy(ref.forward, prompt, past_r, lambda a: np.asarray(a))
        assert tg == tr, f"turn {turn}: generated tokens differ {tg} vs {tr}"
        np.testing.assert_allclose(lg_.cpu().numpy(), lr_, atol=2e-4, rtol=2e-4)
        assert past_g[0][0].shape[2] == past_r[0][0].shape[2]
        for lg, lr in zip(past_g, past_r):
            np.testing.assert_allclose(lg[0].cpu().numpy(), lr[0], atol=1e-5, rtol=1e-5)
    assert pruned_g == pruned_r > 0 and kv_cache.n_pruned_total == pruned_g
    assert "SpAttenKVCache: keep start: 4" in capsys.readouterr().out


def test_multi_turn_protocol_cascade_importance_mode():
    """Extension (parity unpinned): importance = accumulated softmax probabilities; the patched forward accumulates,
    the cache prunes by them and carries the accumulators through the prune.  GPU vs the oracle's restatement."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(1)
    model = TinyLlama().cuda().float()
    for p in model.parameters():
        p.data.mul_(0.6)
    ref = NumpyReplica(model, cascade=True)
    kv_cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, importance_mode="cascade")
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, VOCAB, size=n)[None] for n in (36, 30, 22)]
    past_g = past_r = None
    for turn, prompt in enumerate(prompts):
        if turn > 0:
            space_needed = prompt.shape[1] + MAX_GEN
            Lc = past_r[0][0].shape[2]
            for i in range(L):        # accumulators agree before the prune
                np.testing.assert_allclose(kv_cache.cascade.acc[i][:, :Lc].cpu().numpy(), ref.acc[i][:, :Lc], rtol=2e-3, atol=2e-5)
            past_g = kv_cache.apply_token_pruning(past_g, space_needed, None)
            if Lc + space_needed > START + IMPORTANT + RECENT:
                lo, hi = START, min(Lc - RECENT + space_needed, Lc)
                new_r = []
                for i in range(L):
                    idx = orc.topk_window(ref.acc[i][:, :Lc], lo, hi, IMPORTANT)
                    assert np.array_equal(kv_cache.keep_indices[i].cpu().numpy(), idx), f"turn {turn} layer {i}"
                    Kn, Vn = orc.kv_compact(past_r[i][0], past_r[i][1], idx, START, hi)
                    a = ref.acc[i][:, :Lc]
                    ref.acc[i] = np.concatenate([a[:, :START], np.take_along_axis(a, idx, 1), a[:, hi:]], 1)
                    new_r.append((Kn, Vn))
                past_r = new_r
        tg, past_g, _ = greedy(lambda i, p: model(i, p), torch.from_numpy(prompt).cuda(), past_g,
                               lambda a: torch.tensor(a, device="cuda"))
        tr, past_r, _ = greedy(ref.forward, prompt, past_r, lambda a: np.asarray(a))
        assert tg == tr, f"turn {turn}"
        assert past_g[0][0].shape[2] == past_r[0][0].shape[2]
    assert kv_cache.n_pruned_total > 0
