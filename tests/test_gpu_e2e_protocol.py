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


def acc_of(kv_cache, layer, n):
    """Cumulative importance of a layer INCLUDING the decode step still pending in the fused accumulation."""
    kv_cache.ext.flush(layer)
    return kv_cache.ext.layers[layer].acc[:, :n].cpu().numpy()


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
                np.testing.assert_allclose(acc_of(kv_cache, i, Lc), ref.acc[i][:, :Lc], rtol=2e-3, atol=2e-5)
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


@pytest.mark.parametrize("importance_mode", ["reference", "cascade"])
def test_multi_turn_protocol_global_token_scope(importance_mode):
    """Extension (parity unpinned; README.md:21, workloads/small.csv:1): token_scope="global" — ONE kept token set per layer
    for all heads, ranked by the importance summed over the heads (the last decode step's logits in reference mode, the
    accumulated probabilities in cascade mode).  GPU plugin vs the oracle's restatement over three turns."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(3)
    model = TinyLlama().cuda().float()
    for p in model.parameters():
        p.data.mul_(0.6)
    casc = importance_mode == "cascade"
    ref = NumpyReplica(model, cascade=casc)
    kv_cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, importance_mode=importance_mode, token_scope="global")
    attn_modules = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]
    rng = np.random.default_rng(4)
    prompts = [rng.integers(0, VOCAB, size=n)[None] for n in (38, 28, 24)]
    past_g = past_r = None
    for turn, prompt in enumerate(prompts):
        if turn > 0:
            space_needed = prompt.shape[1] + MAX_GEN
            Lc = past_r[0][0].shape[2]
            scores_r = [ref.acc[i][:, :Lc] for i in range(L)] if casc else [orc.importance(s, "f32") for s in ref.stash]
            past_g = kv_cache.apply_token_pruning(past_g, space_needed, None if casc else [m.attn_scores for m in attn_modules])
            assert Lc + space_needed > START + IMPORTANT + RECENT
            new_r, idxs = orc.global_token_prune(past_r, space_needed, scores_r, START, RECENT, IMPORTANT)
            hi = min(Lc - RECENT + space_needed, Lc)
            for i in range(L):
                got = kv_cache.keep_indices[i].cpu().numpy()
                assert np.array_equal(got, idxs[i]), f"turn {turn} layer {i}"
                assert all(np.array_equal(got[0], got[h]) for h in range(H))
                if casc:      # the per-head accumulators follow the shared row map
                    a = ref.acc[i][:, :Lc]
                    ref.acc[i] = np.concatenate([a[:, :START], np.take_along_axis(a, idxs[i], 1), a[:, hi:]], 1)
            past_r = new_r
        tg, past_g, _ = greedy(lambda i, p: model(i, p), torch.from_numpy(prompt).cuda(), past_g,
                               lambda a: torch.tensor(a, device="cuda"))
        tr, past_r, _ = greedy(ref.forward, prompt, past_r, lambda a: np.asarray(a))
        assert tg == tr, f"turn {turn}"
        assert past_g[0][0].shape[2] == past_r[0][0].shape[2]
    assert kv_cache.n_pruned_total > 0


class ExtReplica(NumpyReplica):
    """NumpyReplica + the oracle's restatements of head pruning / progressive quantisation / local V pruning at
    single-token steps (the modes enable_spatten_llm wires into the patched forward)."""

    def __init__(self, model, cascade=False, head_keep=None, pq_threshold=None, local_v_keep=None, pq_profile=None):
        super().__init__(model, cascade)
        self.head_keep, self.pq_threshold, self.local_v_keep = head_keep, pq_threshold, local_v_keep
        self.pq_profile = pq_profile
        self.head_abs = [np.zeros(H, np.float32) for _ in range(L)]
        self.kept = [None] * L
        self.need = [None] * L

    def select_heads(self):
        if self.head_keep is not None:
            self.kept = orc.head_prune_cascade(self.head_abs, [self.head_keep] * L, self.kept)

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
            qh, kh, vh = sp(x @ w["q_proj"].T), sp(x @ w["k_proj"].T), sp(x @ w["v_proj"].T)
            pk, pv = (None, None) if past is None else past[i]
            if q == 1 and (self.pq_threshold is not None or self.local_v_keep is not None):
                kc, vc = np.concatenate([pk, kh], 2), np.concatenate([pv, vh], 2)
                cos, sin = orc.rope_table(N, D, "f32")
                qr = orc.apply_rotary_pos_emb_single(qh, cos, sin, pos, "f32")[:, :, 0]
                kr = orc.apply_rotary_pos_emb_single(kc, cos, sin, np.arange(N)[None], "f32")
                if self.pq_threshold is not None and self.pq_profile is not None:
                    # bit profile: key MSB plane of kb bits (+ 4 LSBs on refetch), value plane of vb bits, LSB-only refetch
                    kb, vb = self.pq_profile
                    msb, lsb, scale = orc.pq_quantize(kr, bits=kb + 4)
                    qv, vscale = orc.pq_quantize_values(vc, bits=vb)
                    o, self.need[i], logits = orc.pq_decode_attention_profile(qr, msb, lsb, scale, qv, vscale, self.pq_threshold)
                elif self.pq_threshold is not None:
                    msb, lsb, scale = orc.pq_quantize(kr)
                    logits, self.need[i] = orc.pq_logits(qr, msb, lsb, scale, self.pq_threshold)
                    o = np.einsum("bhl,bhld->bhd", orc.softmax_probs(logits), vc)
                else:
                    logits = np.einsum("bhd,bhld->bhl", qr, kr) / np.float32(np.sqrt(D))
                    keep = max(1, min(N, int(np.ceil(self.local_v_keep * N))))
                    o = orc.local_value_prune(orc.softmax_probs(logits), vc, keep)
                o, stash, kv = o.reshape(B, 1, H * D), logits[:, :, None, :], (kc, vc)
            else:
                o, stash, kv = orc.attention_core(qh, kh, vh, pk, pv, pos, mask, "f32")
            if self.kept[i] is not None:                         # pruned heads contribute nothing
                o = o.reshape(B, q, H, D).copy()
                o[:, :, np.setdiff1d(np.arange(H), self.kept[i])] = 0
                o = o.reshape(B, q, H * D)
            self.head_abs[i] = self.head_abs[i] + orc.head_scores(o, H)
            self.stash[i] = stash
            if self.cascade:
                self.accumulate(i, stash, mask)
            x = x + o @ w["o_proj"].T
            new_past.append(kv)
        return x @ self.lm.T, new_past


@pytest.mark.parametrize("mode", ["head", "pq", "local_v", "head+pq+cascade", "pq48", "pq88", "head+pq66"])
def test_multi_turn_protocol_extension_modes(mode):
    """configs[2] / configs[4] through the plugin surface: head pruning, progressive quantisation and local V pruning are
    reached by enable_spatten_llm kwargs and run inside llama_pos_shift_attention_forward + apply_token_pruning; GPU vs
    the oracle replica, multi-turn, fp32 (PARITY UNPINNED: the oracle's restatement of the RTL rules)."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(3)
    model = TinyLlama().cuda().float()
    for p in model.parameters():
        p.data.mul_(0.6)
    kw = {}
    if "head" in mode:
        kw["head_keep"] = 3
    if "pq" in mode:
        kw["pq_threshold"] = 0.06
    for tag, prof in (("pq48", (4, 8)), ("pq88", (8, 8)), ("pq66", (6, 6))):
        if tag in mode:
            kw["pq_profile"] = prof
    if "local_v" in mode:
        kw["local_v_keep"] = 0.4
    cascade = "cascade" in mode
    ref = ExtReplica(model, cascade=cascade, **kw)
    kv_cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, importance_mode="cascade" if cascade else "reference", **kw)
    attn_modules = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]
    rng = np.random.default_rng(4)
    prompts = [rng.integers(0, VOCAB, size=n)[None] for n in (38, 26, 31)]
    past_g = past_r = None
    for turn, prompt in enumerate(prompts):
        if turn > 0:
            space_needed = prompt.shape[1] + MAX_GEN
            Lc = past_r[0][0].shape[2]
            scores = [m.attn_scores for m in attn_modules]
            past_g = kv_cache.apply_token_pruning(past_g, space_needed, scores)
            ref.select_heads()
            if "head" in mode:
                for i in range(L):
                    got = kv_cache.ext.layers[i].head_ids
                    got = np.arange(H) if got is None else got.cpu().numpy()
                    assert np.array_equal(got, ref.kept[i]), f"turn {turn} layer {i}: kept heads"
            assert Lc + space_needed > START + IMPORTANT + RECENT
            lo, hi = START, min(Lc - RECENT + space_needed, Lc)
            new_r = []
            for i in range(L):
                imp = ref.acc[i][:, :Lc] if cascade else orc.importance(ref.stash[i], "f32")
                idx = orc.topk_window(imp, lo, hi, IMPORTANT)
                live = np.arange(H) if ref.kept[i] is None else ref.kept[i]      # a pruned head's rows are dead weight
                assert np.array_equal(kv_cache.keep_indices[i].cpu().numpy()[live], idx[live]), f"turn {turn} layer {i}"
                idx_g = kv_cache.keep_indices[i].cpu().numpy()                     # follow the GPU for dead heads
                Kn, Vn = orc.kv_compact(past_r[i][0], past_r[i][1], idx_g, START, hi)
                if cascade:
                    a = ref.acc[i][:, :Lc]
                    ref.acc[i] = np.concatenate([a[:, :START], np.take_along_axis(a, idx_g, 1), a[:, hi:]], 1)
                new_r.append((Kn, Vn))
            past_r = new_r
        tg, past_g, lg_ = greedy(lambda i, p: model(i, p), torch.from_numpy(prompt).cuda(), past_g,
                                 lambda a: torch.tensor(a, device="cuda"))
        tr, past_r, lr_ = greedy(ref.forward, prompt, past_r, lambda a: np.asarray(a))
        assert tg == tr, f"turn {turn}: generated tokens differ {tg} vs {tr}"
        np.testing.assert_allclose(lg_.cpu().numpy(), lr_, atol=5e-4, rtol=5e-4)
        if "pq" in mode:
            for i in range(L):
                need_g = kv_cache.ext.layers[i].need_lsb.cpu().numpy().reshape(1, H).astype(bool)
                live = np.arange(H) if ref.kept[i] is None else ref.kept[i]
                assert np.array_equal(need_g[:, live], ref.need[i][:, live]), f"turn {turn} layer {i}: refetch flags"
    assert kv_cache.n_pruned_total > 0
    if "head" in mode:
        assert all(k is not None and len(k) == 3 for k in ref.kept)


def test_multi_turn_protocol_layer_cascade():
    """Layer-to-layer cascade token pruning through the plugin (enable_spatten_llm(layer_keep=[...])): the token set
    shrinks from layer to layer, the caches of the layers get different lengths, every layer rotates with its own
    cache-relative positions.  GPU vs the oracle's restatement (PARITY UNPINNED), multi-turn, fp32."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(5)
    model = TinyLlama().cuda().float()
    for p in model.parameters():
        p.data.mul_(0.6)
    keeps = [IMPORTANT, IMPORTANT - 8]
    ref = NumpyReplica(model)
    kv_cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, layer_keep=keeps)
    attn_modules = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]

    def ref_forward(ids, past):
        """The replica with PER-LAYER cache lengths (positions / causal mask relative to each layer's own cache)."""
        B, q = ids.shape
        x = ref.emb[ids]
        new_past = []
        sp = lambda t: np.swapaxes(t.reshape(B, q, H, D), 1, 2)
        for i, w in enumerate(ref.w):
            P = 0 if past is None else past[i][0].shape[2]
            pos = np.tile(np.arange(P, P + q)[None], (B, 1))
            o, stash, kv = orc.attention_core(sp(x @ w["q_proj"].T), sp(x @ w["k_proj"].T), sp(x @ w["v_proj"].T),
                                              None if past is None else past[i][0], None if past is None else past[i][1],
                                              pos, orc.causal_mask(B, q, P + q, "f32"), "f32")
            ref.stash[i] = stash
            x = x + o @ w["o_proj"].T
            new_past.append(kv)
        return x @ ref.lm.T, new_past

    rng = np.random.default_rng(6)
    prompts = [rng.integers(0, VOCAB, size=n)[None] for n in (40, 24, 28, 16)]
    past_g = past_r = None
    ids_r, next_id = None, 0
    for turn, prompt in enumerate(prompts):
        if turn > 0:
            space_needed = prompt.shape[1] + MAX_GEN
            scores = [m.attn_scores for m in attn_modules]
            lens_before = [kv[0].shape[2] for kv in past_r]
            past_g = kv_cache.apply_token_pruning(past_g, space_needed, scores)
            # token ids on the replica side: known ids + fresh ones for the rows appended since the last prune
            n_new = lens_before[0] - (0 if ids_r is None else ids_r[0].shape[1])
            fresh = np.tile(np.arange(next_id, next_id + n_new, dtype=np.int32)[None], (H, 1))
            ids_now = [fresh if ids_r is None else np.concatenate([ids_r[i], fresh], 1) for i in range(L)]
            next_id += n_new
            imp = [orc.importance(ref.stash[i], "f32") for i in range(L)]
            past_r, ids_r, idxs = orc.layer_cascade_prune(past_r, ids_now, imp, space_needed, START, RECENT, keeps)
            for i in range(L):
                assert np.array_equal(kv_cache.keep_indices[i].cpu().numpy(), idxs[i]), f"turn {turn} layer {i}"
                assert past_g[i][0].shape[2] == past_r[i][0].shape[2] == START + keeps[i] + (lens_before[i] - min(lens_before[i] - RECENT + space_needed, lens_before[i]))
                np.testing.assert_allclose(past_g[i][0].cpu().numpy(), past_r[i][0], atol=1e-5, rtol=1e-5)
                assert np.array_equal(kv_cache.ext.tok_ids[i].cpu().numpy(), ids_r[i])
            # nesting: what layer 1 kept of its window is a subset of what layer 0 kept (per head)
            for h in range(H):
                assert set(ids_r[1][h].tolist()) <= set(ids_r[0][h].tolist())
            assert past_g[1][0].shape[2] < past_g[0][0].shape[2]
        tg, past_g, lg_ = greedy(lambda i, p: model(i, p), torch.from_numpy(prompt).cuda(), past_g,
                                 lambda a: torch.tensor(a, device="cuda"))
        tr, past_r, lr_ = greedy(ref_forward, prompt, past_r, lambda a: np.asarray(a))
        assert tg == tr, f"turn {turn}: generated tokens differ {tg} vs {tr}"
        np.testing.assert_allclose(lg_.cpu().numpy(), lr_, atol=5e-4, rtol=5e-4)
    assert kv_cache.n_pruned_total > 0


def test_extension_modes_refuse_a_non_causal_mask():
    """Round-2 advisor finding: with an extension mode on, the forward used to drop HF's attention_mask silently — a
    left-padded batch then attended to its padding.  It now refuses any multi-token mask that is not the causal one."""
    from spatten_amd import enable_spatten_llm
    torch.manual_seed(0)
    model = TinyLlama().cuda().float()
    enable_spatten_llm(model, start_size=START, important_size=IMPORTANT, recent_size=RECENT, importance_mode="cascade")
    attn = model.layers[0].self_attn
    B, q = 2, 12
    x = torch.randn(B, q, HID, device="cuda")
    pos = torch.arange(q, device="cuda")[None].expand(B, q)
    causal = torch.from_numpy(orc.causal_mask(B, q, q, "f32")).cuda()
    attn(x, attention_mask=causal, position_ids=pos, past_key_value=None, use_cache=True)       # the HF causal mask: fine
    padded = causal.clone()
    padded[1, :, :, :3] = torch.finfo(torch.float32).min                                         # left padding of sequence 1
    with pytest.raises(ValueError, match="causal mask"):
        attn(x, attention_mask=padded, position_ids=pos, past_key_value=None, use_cache=True)
