"""Prefill attention (rows leg + MFMA flash leg) and the patched forward vs reference goldens / oracle.  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, TORCH_DT, attn_inputs, check_stash, dev, golden, host

pytestmark = pytest.mark.gpu


def run_prefill(q, k, v, past, dt, mask=None, causal=False, stash=True, colimp=False, table="oracle", pos_t=None,
                numerics="reference"):
    from spatten_amd import ops
    B, H, ql, d = q.shape
    Hkv = k.shape[1]
    P = 0 if past is None else past[0].shape[2]
    N = P + ql
    kc = k if past is None else np.concatenate([past[0], k], 2)
    vc = v if past is None else np.concatenate([past[1], v], 2)
    cap = N + 7
    kd = torch.full((B, Hkv, cap, d), float("nan"), dtype=TORCH_DT[dt], device="cuda")
    vd = torch.full((B, Hkv, cap, d), float("nan"), dtype=TORCH_DT[dt], device="cuda")
    kd[:, :, :N] = dev(kc, dt)
    vd[:, :, :N] = dev(vc, dt)
    if table == "torch":
        cos, sin = ops.rope_table(N + 3, d, TORCH_DT[dt], "cuda")
    else:
        c, s = orc.rope_table(N + 3, d, dt)
        cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    krd = torch.full_like(kd, float("nan"))
    ops.build_shadow(kd, krd, 0, N, cos, sin)
    scores = torch.full((B, H, ql, N), float("nan"), dtype=TORCH_DT[dt], device="cuda") if stash else None
    ci = torch.zeros(B, H, N, dtype=torch.float32, device="cuda") if colimp else None
    # q in the projection layout [B,q,H*d] viewed as [B,H,q,d] (no copy), like the forward passes it
    qd = dev(np.swapaxes(q, 1, 2).reshape(B, ql, H * d), dt).view(B, ql, H, d).transpose(1, 2)
    out = ops.attn_prefill(qd, krd, vd, N, cos, sin, P, causal=causal, position_ids=pos_t,
                           mask=None if mask is None else dev(mask, dt), scores=scores, col_importance=ci, numerics=numerics)
    torch.cuda.synchronize()
    return host(out), (None if scores is None else host(scores)), (None if ci is None else host(ci))


def test_prefill_matches_reference_goldens():
    g = golden("g3_attention.npz")
    n = 0
    for m in g["meta"]:
        name, B, H, Hkv, d, P, ql, mask_kind, dt, seed = m.split("|")
        B, H, Hkv, d, P, ql, seed = map(int, (B, H, Hkv, d, P, ql, seed))
        if ql == 1:
            continue
        q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed)
        N = P + ql
        mask = orc.causal_mask(B, ql, N, dt)[:, 0] if mask_kind == "causal" else None
        out, stash, _ = run_prefill(q, k, v, past, dt, mask=mask, table="torch")          # explicit HF mask
        np.testing.assert_allclose(out, g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
        check_stash(stash, g[f"{name}_stash"], dt, name)
        if mask_kind == "causal":                                                          # causal flag, no mask read
            out2, stash2, _ = run_prefill(q, k, v, past, dt, causal=True, table="torch")
            np.testing.assert_allclose(out2, g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
            check_stash(stash2, g[f"{name}_stash"], dt, name)
            out3, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False, table="torch")   # tile skipping
            np.testing.assert_allclose(out3, g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
        n += 1
    assert n >= 7


@pytest.mark.parametrize("dt,d", [("bf16", 128), ("f16", 64), ("bf16", 64), ("f32", 128)])
@pytest.mark.parametrize("P,ql", [(0, 130), (300, 200), (77, 9), (513, 64)])
def test_prefill_vs_oracle(dt, d, P, ql):
    B, H, Hkv = 2, 4, 2
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=300 + P + ql)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    mask = orc.causal_mask(B, ql, N, dt)
    o, stash, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos, mask, dt)
    out, st, ci = run_prefill(q, k, v, past, dt, causal=True, colimp=(dt != "f32" and ql > 8))
    np.testing.assert_allclose(out, o, **OUT_TOL[dt])
    check_stash(st, stash, dt)
    if ci is not None:       # reference-mode importance without the stash: sum over query rows, acausal logits included
        np.testing.assert_allclose(ci, st.sum(axis=2, dtype=np.float32), rtol=1e-4, atol=1e-3)
    # explicit arbitrary mask + explicit position ids (shifted), no causal flag
    rng = np.random.default_rng(1)
    am = np.where(rng.random((B, 1, ql, N)) < 0.2, np.float32(orc.finfo_min(dt)), np.float32(0)).astype(np.float32)
    am[..., 0] = 0
    pos2 = pos + 3
    cos, sin = orc.rope_table(N + 3, d, dt)
    qr = orc.apply_rotary_pos_emb_single(q, cos, sin, pos2, dt)
    kc = k if past is None else np.concatenate([past[0], k], 2)
    vc = v if past is None else np.concatenate([past[1], v], 2)
    kr = orc.repeat_kv(orc.apply_rotary_pos_emb_single(kc, cos, sin, np.arange(N)[None], dt), H // Hkv)
    s = orc.round_dt(orc.round_dt(np.matmul(qr, np.swapaxes(kr, 2, 3)), dt) / np.float32(np.sqrt(d)), dt)
    pm = orc.softmax_probs(orc.round_dt(s + am, dt))
    o2 = np.swapaxes(np.matmul(pm, orc.repeat_kv(vc, H // Hkv)), 1, 2).reshape(B, ql, H * d)
    out2, st2, _ = run_prefill(q, k, v, past, dt, mask=am[:, 0], pos_t=torch.from_numpy(pos2).cuda())
    np.testing.assert_allclose(out2, orc.round_dt(o2, dt), **OUT_TOL[dt])
    check_stash(st2, s, dt)


@pytest.mark.parametrize("case", [("bf16", 128, 1, 4, 4, 0, 700), ("bf16", 128, 2, 4, 1, 411, 300), ("f16", 128, 1, 2, 2, 1000, 257),
                                  ("bf16", 64, 1, 8, 2, 129, 520), ("f16", 64, 2, 2, 2, 0, 385), ("bf16", 128, 1, 2, 2, 255, 1),
                                  # whole 256-row groups, causal: the PAIRED form (a workgroup's halves take the 128-row blocks
                                  # i and n - 1 - i) — with and without a cache in front, GQA, batch, d = 64, one and three pairs
                                  ("bf16", 128, 1, 4, 4, 0, 1024), ("bf16", 128, 2, 4, 2, 300, 768), ("f16", 64, 1, 2, 2, 77, 512),
                                  ("f16", 128, 1, 2, 1, 1000, 1536)])
def test_prefill_multi_block_shapes_vs_oracle(case):
    """Several 256-query workgroups per head, ragged last blocks, tiles that straddle N, GQA, batch: the by-product-free
    kernel (128-key tiles, LDS-DMA; paired 128-row blocks where the block is whole 256-row groups) and the stash kernel
    (64-key tiles) against the oracle."""
    dt, d, B, H, Hkv, P, ql = case
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=900 + P + ql)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    o, stash, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos,
                                     orc.causal_mask(B, ql, N, dt), dt)
    out, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False)
    np.testing.assert_allclose(out, o, **OUT_TOL[dt])
    out, st, _ = run_prefill(q, k, v, past, dt, causal=True, stash=True)
    np.testing.assert_allclose(out, o, **OUT_TOL[dt])
    check_stash(st, stash, dt)
    # non-causal (every key visible to every query), no mask
    o2, _, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos, None, dt)
    out2, _, _ = run_prefill(q, k, v, past, dt, causal=False, stash=False)
    np.testing.assert_allclose(out2, o2, **OUT_TOL[dt])


@pytest.mark.parametrize("case", [("bf16", 128, 1, 8, 8, 2000, 40), ("bf16", 128, 1, 8, 2, 1500, 300), ("f16", 64, 2, 4, 4, 900, 17),
                                  ("bf16", 128, 1, 4, 4, 0, 700)])
def test_turn_prefill_key_split_vs_oracle(case, monkeypatch):
    """A short block of new tokens on a long cache (the turn prefill of the multi-turn protocol): the flash kernel splits
    the KEY tiles of a query block over several workgroups and a merge kernel folds the partials — outputs and the row
    statistics vs the oracle, causal flag and explicit mask, and the same call with the split forced off."""
    from spatten_amd import ops
    dt, d, B, H, Hkv, P, ql = case
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=1300 + P + ql)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    pk, pv = (None, None) if past is None else past
    o, stash, _ = orc.attention_core(q, k, v, pk, pv, pos, orc.causal_mask(B, ql, N, dt), dt)
    out, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False)
    np.testing.assert_allclose(out, o, **OUT_TOL[dt])
    # row statistics through the split: (m, l) of every row = max / sum-exp of the oracle's masked logits
    kc = k if past is None else np.concatenate([pk, k], 2)
    vc = v if past is None else np.concatenate([pv, v], 2)
    cos, sin = ops.rope_table(N + 3, d, TORCH_DT[dt], "cuda")
    kd, vd = dev(kc, dt), dev(vc, dt)
    krd = ops.rope_single(kd, cos, sin)
    qd = dev(q, dt)
    lse = torch.empty(B, H, ql, 2, dtype=torch.float32, device="cuda")
    out2 = ops.attn_prefill(qd, krd, vd, N, cos, sin, P, causal=True, lse=lse)
    logits = stash.astype(np.float32) + orc.causal_mask(B, ql, N, dt).astype(np.float32)
    m_ref = logits.max(-1)
    l_ref = np.exp(logits - m_ref[..., None]).sum(-1)
    got = host(lse)
    # the kernel's m may lag the true maximum (deferred rescale): compare the invariant m + log l
    np.testing.assert_allclose(got[..., 0] + np.log(got[..., 1]), m_ref + np.log(l_ref), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(host(out2), o, **OUT_TOL[dt])
    # explicit (non-causal) mask through the same split
    rng = np.random.default_rng(5)
    am = np.where(rng.random((B, 1, ql, N)) < 0.3, np.float32(orc.finfo_min(dt)), np.float32(0)).astype(np.float32)
    am[..., 0] = 0
    o3, _, _ = orc.attention_core(q, k, v, pk, pv, pos, am, dt)
    out3, _, _ = run_prefill(q, k, v, past, dt, mask=am[:, 0], stash=False)
    np.testing.assert_allclose(out3, o3, **OUT_TOL[dt])


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_prefill_deferred_rescale_spikes(dt):
    """The no-stash flash path only moves its running maximum when a row outgrows it by more than a threshold.  Rare
    and data dependent on random inputs, so force it: keys whose logits jump by > threshold (must rescale), by less
    (deferred: P > 1 until the next move) and again by more, at different tiles, with per-row magnitudes."""
    B, H, Hkv, d, P, ql = 1, 2, 2, 128, 768, 256
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=4242)
    N = P + ql
    rng = np.random.default_rng(7)
    # logit ~ a_q * b_j * 2 / sqrt(d) on the slowest rotary pair (dims 63 / 127: ~identity under RoPE at these positions)
    a = orc.round_dt((6.0 * (0.5 + rng.random((B, H, ql)))).astype(np.float32), dt)
    q[..., 63] = a
    q[..., 127] = a
    kc = np.concatenate([past[0], k], 2)
    kc[..., 63] = 0
    kc[..., 127] = 0
    for pos, b in ((40, 4.0), (300, 9.0), (430, 11.0), (700, 22.0), (850, 23.5)):   # tiles 0, 2, 3, 5, 6 of 128 keys
        kc[:, :, pos, 63] = b
        kc[:, :, pos, 127] = b
    kc = orc.round_dt(kc, dt)
    past = (kc[:, :, :P], past[1])
    k = kc[:, :, P:]
    pos_ids = np.tile(np.arange(P, N)[None], (B, 1))
    o, stash, _ = orc.attention_core(q, k, v, past[0], past[1], pos_ids, orc.causal_mask(B, ql, N, dt), dt)
    assert np.diff(np.sort(stash[0, 0, 0, [40, 300, 430, 700]])).max() > 8.0        # the jumps are really there
    for variant_stash in (False, True):
        out, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=variant_stash)
        np.testing.assert_allclose(out, o, **OUT_TOL[dt])


# ---- the patched forward through a stub module with the transformers-4.33 attribute surface ---------
class Fixed(nn.Module):
    def __init__(self):
        super().__init__()
        self.t = None

    def forward(self, x):
        return self.t


class StubAttn(nn.Module):
    _spatten_llama_attention = True

    def __init__(self, H, Hkv, d):
        super().__init__()
        self.config = SimpleNamespace(pretraining_tp=1, model_type="llama")
        self.num_heads, self.num_key_value_heads, self.head_dim = H, Hkv, d
        self.num_key_value_groups = H // Hkv
        self.hidden_size = H * d
        self.q_proj, self.k_proj, self.v_proj = Fixed(), Fixed(), Fixed()
        self.o_proj = nn.Identity()


def fwd(m, q, k, v, past, pos, mask, dt):
    from spatten_amd.pos_shift.modify_llama import llama_pos_shift_attention_forward
    B, H, ql, d = q.shape
    Hkv = k.shape[1]
    m.q_proj.t = dev(np.swapaxes(q, 1, 2).reshape(B, ql, H * d), dt)
    m.k_proj.t = dev(np.swapaxes(k, 1, 2).reshape(B, ql, Hkv * d), dt)
    m.v_proj.t = dev(np.swapaxes(v, 1, 2).reshape(B, ql, Hkv * d), dt)
    hidden = torch.zeros(B, ql, H * d, dtype=TORCH_DT[dt], device="cuda")
    return llama_pos_shift_attention_forward(m, hidden, attention_mask=mask, position_ids=pos, past_key_value=past,
                                             use_cache=True)


def test_forward_matches_reference_goldens():
    g = golden("g3_attention.npz")
    for mline in g["meta"]:
        name, B, H, Hkv, d, P, ql, mask_kind, dt, seed = mline.split("|")
        B, H, Hkv, d, P, ql, seed = map(int, (B, H, Hkv, d, P, ql, seed))
        q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed)
        N = P + ql
        m = StubAttn(H, Hkv, d)
        pos = torch.arange(P, N, device="cuda")[None].expand(B, ql)
        mask = None
        if mask_kind == "zeros":
            mask = torch.zeros(B, 1, ql, N, dtype=TORCH_DT[dt], device="cuda")
        elif mask_kind == "causal":
            mask = dev(orc.causal_mask(B, ql, N, dt), dt)
        pkv = None if past is None else (dev(past[0], dt), dev(past[1], dt))
        out, w, new_past = fwd(m, q, k, v, pkv, pos, mask, dt)
        torch.cuda.synchronize()
        assert w is None and out.shape == (B, ql, H * d)
        np.testing.assert_allclose(host(out), g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
        check_stash(host(m.attn_scores), g[f"{name}_stash"], dt, name)
        assert m.attn_scores.shape == (B, H, ql, N)
        # returned cache: un-rotated concat, bit exact, HF layout
        want_k = k if past is None else np.concatenate([past[0], k], 2)
        assert new_past[0].shape == (B, Hkv, N, d) and np.array_equal(host(new_past[0]), want_k), name
        if pkv is not None:
            assert np.array_equal(host(pkv[0]), past[0])       # inputs never mutated
        # extension flags (enable_spatten_llm(..., prefill_stash=False, assume_causal=True)): same output, no stash for
        # multi-token forwards, the decode stash still written
        if mask_kind == "causal":
            m2 = StubAttn(H, Hkv, d)
            m2.spatten_prefill_stash, m2.spatten_assume_causal = False, True
            out2, _, _ = fwd(m2, q, k, v, pkv, pos, mask, dt)
            torch.cuda.synchronize()
            np.testing.assert_allclose(host(out2), g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
            assert (m2.attn_scores is None) == (ql > 1)
            # default flags: the HF causal mask is recognised (once per mask tensor) and the causal kernel path is taken
            from spatten_amd.pos_shift.modify_llama import _mask_is_causal
            if ql > 1:
                assert _mask_is_causal(mask, P) and mask._spatten_is_causal[1] is True
                bad = mask.clone()
                bad[0, 0, ql - 1, 0] = torch.finfo(bad.dtype).min          # a padding-style hole: not purely causal
                assert not _mask_is_causal(bad, P)


@pytest.mark.parametrize("assume", [False, True])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_forward_multi_step_decode_and_prune_roundtrip(dt, assume):
    """prefill -> 5 decode steps (in-place appends into the slab) -> prune -> decode again: every step vs the oracle
    run on the reference's semantics (cat + re-rotate everything each step).  ``assume``: the assume_causal extension —
    single-token steps then ignore the HF mask / position_ids (lean decode kernel); either way they go through the slab's
    prefilled argument block."""
    from spatten_amd import SpAttenKVCache
    B, H, d = 1, 4, 128
    m = StubAttn(H, H, d)
    m.spatten_assume_causal = assume
    rng_seed = 77
    past_np, past_dev = None, None
    L = 0
    for step, ql in enumerate([150, 1, 1, 1, 1, 1]):
        q = orc.synth_normal(rng_seed, 10 * step, (B, H, ql, d), dt)
        k = orc.synth_normal(rng_seed, 10 * step + 1, (B, H, ql, d), dt)
        v = orc.synth_normal(rng_seed, 10 * step + 2, (B, H, ql, d), dt)
        N = L + ql
        pos = np.tile(np.arange(L, N)[None], (B, 1))
        mask_np = orc.causal_mask(B, ql, N, dt)
        o, stash, (kc, vc) = orc.attention_core(q, k, v, None if past_np is None else past_np[0],
                                                None if past_np is None else past_np[1], pos, mask_np, dt)
        out, _, past_dev = fwd(m, q, k, v, past_dev, torch.from_numpy(pos).cuda(), dev(mask_np, dt), dt)
        # the oracle's table is numpy's; the forward's is torch's: same values except rare 1-ulp table entries,
        # so compare with the output tolerance only
        np.testing.assert_allclose(host(out), o, err_msg=f"step {step}", **OUT_TOL[dt])
        assert np.array_equal(host(past_dev[0]), kc) and np.array_equal(host(past_dev[1]), vc)
        if ql == 1:
            check_stash(host(m.attn_scores), stash, dt, f"step {step}")
            assert past_dev[0]._spatten_slab.dec is not None        # the slab's prefilled argument block served the step
        past_np, L = (kc, vc), N
    slab = past_dev[0]._spatten_slab
    assert slab.length == L and slab.rot_len == L and slab.capacity >= L
    # prune with the last stash, then one more decode step on the pruned cache
    cache = SpAttenKVCache(start_size=4, recent_size=40, important_size=60)
    stash_dev = m.attn_scores
    pruned = cache.apply_token_pruning([past_dev], 10, [stash_dev])
    want, idx = orc.apply_token_pruning([past_np], 10, [host(stash_dev)], 4, 40, 60, dt)
    assert np.array_equal(host(pruned[0][0]), want[0][0]) and np.array_equal(host(pruned[0][1]), want[0][1])
    Lp = want[0][0].shape[2]
    q = orc.synth_normal(rng_seed, 100, (B, H, 1, d), dt)
    k = orc.synth_normal(rng_seed, 101, (B, H, 1, d), dt)
    v = orc.synth_normal(rng_seed, 102, (B, H, 1, d), dt)
    pos = np.full((B, 1), Lp)
    o, _, (kc, vc) = orc.attention_core(q, k, v, want[0][0], want[0][1], pos, None, dt)
    out, _, past2 = fwd(m, q, k, v, tuple(pruned[0]), torch.from_numpy(pos).cuda(), None, dt)
    np.testing.assert_allclose(host(out), o, **OUT_TOL[dt])
    assert np.array_equal(host(past2[0]), kc)
    # the pruned slab came with its rotated shadow and spare room: the append was in place
    assert past2[0].data_ptr() == pruned[0][0].data_ptr()


def test_enable_patches_every_llama_attention_and_rejects_others():
    from spatten_amd import enable_spatten_llm
    from spatten_amd.pos_shift.modify_llama import llama_pos_shift_attention_forward

    class Layer(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn = StubAttn(4, 4, 64)
            self.mlp = nn.Linear(4, 4)

    class Model(nn.Module):
        def __init__(self, mt):
            super().__init__()
            self.config = SimpleNamespace(model_type=mt)
            self.layers = nn.ModuleList([Layer(), Layer(), Layer()])

    model = Model("llama")
    cache = enable_spatten_llm(model, start_size=4, important_size=10, recent_size=12)
    assert cache.cache_size == 26
    for layer in model.layers:
        assert layer.self_attn.forward.__func__ is llama_pos_shift_attention_forward
    with pytest.raises(ValueError, match="got gpt2"):
        enable_spatten_llm(Model("gpt2"), 4, 10, 12)


@pytest.mark.parametrize("Hkv,bias", [(4, False), (2, True)])
def test_fused_qkv_projection_matches_the_three_linears(Hkv, bias):
    """enable_spatten_llm(..., fuse_qkv=True): one GEMM over the stacked weights; prefill, then decode steps whose query /
    key / value rows are SLICES of one row — against the same module run with its three nn.Linear projections."""
    import copy
    from spatten_amd import enable_spatten_llm
    dt, tdt, B, H, d, P = "bf16", torch.bfloat16, 1, 4, 64, 300
    torch.manual_seed(5)

    class Attn(nn.Module):
        _spatten_llama_attention = True

        def __init__(self):
            super().__init__()
            hid = H * d
            self.config = SimpleNamespace(pretraining_tp=1, model_type="llama")
            self.num_heads, self.num_key_value_heads, self.head_dim, self.hidden_size = H, Hkv, d, hid
            self.num_key_value_groups = H // Hkv
            self.q_proj = nn.Linear(hid, H * d, bias=bias, dtype=tdt, device="cuda")
            self.k_proj = nn.Linear(hid, Hkv * d, bias=bias, dtype=tdt, device="cuda")
            self.v_proj = nn.Linear(hid, Hkv * d, bias=bias, dtype=tdt, device="cuda")
            self.o_proj = nn.Linear(H * d, hid, bias=False, dtype=tdt, device="cuda")

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(model_type="llama")
            self.layers = nn.ModuleList([Attn()])

    ref = Model()
    fused = copy.deepcopy(ref)
    enable_spatten_llm(ref, 4, 100, 100)
    enable_spatten_llm(fused, 4, 100, 100, fuse_qkv=True)
    mf = fused.layers[0]
    assert mf._spatten_qkv is not None and mf.q_proj.weight.data_ptr() == mf._spatten_qkv[0].data_ptr()   # views, no copy
    assert torch.equal(mf.k_proj.weight, ref.layers[0].k_proj.weight)
    x = torch.randn(B, P, H * d, device="cuda").to(tdt)
    mask = dev(orc.causal_mask(B, P, P, dt), dt)
    pos = torch.arange(P, device="cuda")[None]
    with torch.no_grad():
        outs = []
        for model in (ref, fused):
            m = model.layers[0]
            o, _, past = m(x, attention_mask=mask, position_ids=pos, past_key_value=None, use_cache=True)
            steps = [o]
            for t in range(3):
                n = past[0].shape[2]
                xt = x[:, t:t + 1] * 0.5
                o1, _, past = m(xt, attention_mask=torch.zeros(B, 1, 1, n + 1, dtype=tdt, device="cuda"),
                                position_ids=torch.full((B, 1), n, device="cuda"), past_key_value=past, use_cache=True)
                steps += [o1, m.attn_scores]
            outs.append((steps, past))
    for a_, b_ in zip(outs[0][0], outs[1][0]):
        np.testing.assert_allclose(host(b_), host(a_), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(host(outs[1][1][0]), host(outs[0][1][0]), rtol=2e-2, atol=2e-2)   # the K cache


def test_rope_base_of_the_module_survives_the_prune():
    """Round-1 advisor finding: the prune rebuilt the rotated shadow with base 10000 whatever the model's rotary base is
    (CodeLlama: rope_theta 1e6).  prefill -> decode -> prune -> decode on a module with base 1e6, every step vs the oracle
    run with the same base; and a cache that lost its slab is re-rotated (counted, warned about), not mis-rotated."""
    import warnings
    from spatten_amd import SpAttenKVCache, kv_slab
    dt, B, H, d, ql = "f32", 1, 4, 64, 40
    base = 1.0e6
    m = StubAttn(H, H, d)
    m.rotary_emb = SimpleNamespace(base=base)
    q, k, v, _ = attn_inputs(B, H, H, d, 0, ql, dt, 71)
    pos = torch.arange(0, ql, device="cuda")[None]
    mask = dev(orc.causal_mask(B, ql, ql, dt), dt)
    out, _, past = fwd(m, q, k, v, None, pos, mask, dt)
    o_ref, st_ref, (pk, pv) = orc.attention_core(q, k, v, None, None, np.arange(ql)[None], orc.causal_mask(B, ql, ql, dt), dt, base=base)
    np.testing.assert_allclose(host(out), o_ref, **OUT_TOL[dt])
    assert past[0]._spatten_slab.base == base

    def step(past_g, past_r, seed):
        P = past_r[0].shape[2]
        q1, k1, v1, _ = attn_inputs(B, H, H, d, 0, 1, dt, seed)
        og, _, new_g = fwd(m, q1, k1, v1, past_g, torch.full((B, 1), P, device="cuda"), torch.zeros(B, 1, 1, P + 1, device="cuda"), dt)
        orf, st, new_r = orc.attention_core(q1, k1, v1, past_r[0], past_r[1], np.full((B, 1), P), None, dt, base=base)
        np.testing.assert_allclose(host(og), orf, **OUT_TOL[dt])
        check_stash(host(m.attn_scores), st, dt)
        return new_g, new_r, st

    past_g, past_r, st = step(past, (pk, pv), 72)
    cache = SpAttenKVCache(4, 8, 12)
    past_g = cache.apply_token_pruning([past_g], 6, [m.attn_scores])[0]
    new_r, _ = orc.apply_token_pruning([past_r], 6, [st], 4, 8, 12, dt)
    assert past_g[0]._spatten_slab.base == base                     # the table parameters travel with the slab
    assert np.array_equal(host(past_g[0]), new_r[0][0])
    past_g, past_r, _ = step(tuple(past_g), tuple(new_r[0]), 73)     # wrong tables in the shadow would show here
    # a cache that lost its slab: correct result, one counted re-copy, one warning
    before = kv_slab.recopy_events
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step((past_g[0].clone(), past_g[1].clone()), past_r, 74)
    assert kv_slab.recopy_events == before + 1
    assert before > 0 or any("without their KV slab" in str(x.message) for x in w)


def test_fp32_rows_leg_slices_long_blocks():
    """Round-1 advisor finding: the exact fp32 leg launched one grid-z entry per query row and failed beyond 65535 rows
    (B * q_len).  It now runs in slices of <= 4096 rows; a block longer than one slice (stash, row statistics and causal
    visibility of the later slices included) vs a direct fp32 evaluation on the device."""
    from spatten_amd import ops
    B, H, d, P, ql = 1, 2, 64, 300, 4200
    N = P + ql
    g = torch.Generator(device="cuda").manual_seed(3)
    Q = torch.randn(B, H, ql, d, device="cuda", generator=g)
    K = torch.randn(B, H, N, d, device="cuda", generator=g)
    V = torch.randn(B, H, N, d, device="cuda", generator=g)
    cos, sin = ops.rope_table(N, d, torch.float32, "cuda")
    Kr = ops.rope_single(K, cos, sin)
    Qr = ops.rope_single(Q, cos, sin, pos0=P)
    stash = torch.empty(B, H, ql, N, device="cuda")
    lse = torch.empty(B, H, ql, 2, device="cuda")
    out = ops.attn_prefill(Q, Kr, V, N, cos, sin, P, causal=True, scores=stash, lse=lse)
    s = (Qr @ Kr.transpose(2, 3)) / d ** 0.5
    assert torch.allclose(stash, s, atol=2e-5, rtol=1e-5)                       # the stash is pre-mask: all N columns
    vis = torch.arange(N, device="cuda")[None, :] <= (P + torch.arange(ql, device="cuda"))[:, None]
    sm = s.masked_fill(~vis, float("-inf"))
    want = (torch.softmax(sm, -1) @ V).transpose(1, 2).reshape(B, ql, H * d)
    assert torch.allclose(out, want, atol=2e-5, rtol=1e-4)
    p_from_lse = torch.exp(sm[0, 1, 4150] - lse[0, 1, 4150, 0]) / lse[0, 1, 4150, 1]     # a row of the second slice
    assert torch.allclose(p_from_lse, torch.softmax(sm[0, 1, 4150], -1), atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("dt,d,P,ql", [("bf16", 128, 0, 700), ("f16", 64, 300, 520), ("bf16", 128, 1000, 257)])
def test_prefill_fast_numerics_stays_within_the_stated_tolerance(dt, d, P, ql):
    """numerics="fast" (SPATTEN_PREFILL_FAST_NUMERICS): fp32 logits instead of the reference's two 16-bit roundings.  On
    unit-variance logits the output stays inside the reference tolerance (its divergence at LARGE logits is the documented
    reason it is opt-in: DESIGN §3.4); with a stash requested the roundings come back (the stash is defined on them)."""
    B, H, Hkv = 1, 4, 4
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=900 + P + ql)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    o_ref, stash_ref, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1],
                                             pos, orc.causal_mask(B, ql, N, dt), dt)
    o_fast, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False, numerics="fast")
    o_exact, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False)
    np.testing.assert_allclose(o_fast, o_ref, **OUT_TOL[dt])
    np.testing.assert_allclose(o_exact, o_ref, **OUT_TOL[dt])
    assert not np.array_equal(o_fast, o_exact), "the fast path is expected to be a different instantiation"
    o_st, stash, _ = run_prefill(q, k, v, past, dt, causal=True, stash=True, numerics="fast")
    check_stash(stash, stash_ref, dt, "fast numerics with a stash")
    with pytest.raises(ValueError):
        run_prefill(q, k, v, past, dt, causal=True, stash=False, numerics="sloppy")
    # round 6: numerics="auto" (the default of ops.attn_prefill and of the plugin) = fast exactly when nothing defined on the
    # rounded logits is requested, reference otherwise — bit for bit the explicit forms
    o_auto, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False, numerics="auto")
    assert np.array_equal(o_auto, o_fast)
    o_auto_st, stash_auto, _ = run_prefill(q, k, v, past, dt, causal=True, stash=True, numerics="auto")
    o_ref_st, stash_r, _ = run_prefill(q, k, v, past, dt, causal=True, stash=True, numerics="reference")
    assert np.array_equal(o_auto_st, o_ref_st) and np.array_equal(stash_auto, stash_r)


@pytest.mark.parametrize("case", [("bf16", 1, 8, 8, 2000, 64), ("f16", 2, 4, 2, 700, 300), ("bf16", 1, 4, 4, 0, 512), ("bf16", 1, 4, 1, 3000, 17),
                                  ("bf16", 1, 4, 4, 0, 2304), ("f16", 1, 2, 2, 0, 1300)])
def test_prefill_transposing_read_form_vs_oracle(case):
    """Query blocks of up to 512 rows — and, since round 5, whole prompts (q = N) of up to 2560 tokens — at d = 128 run the flash
    kernel that reads V through gfx950's transposing LDS reads (ds_read_b64_tr_b16) from the value rows themselves — no
    key-contiguous copy of V is made; longer blocks keep the Vt pre-pass.  Both forms against the oracle (GQA, key split,
    ragged lengths, a past longer than the block)."""
    dt, B, H, Hkv, P, ql = case
    d = 128
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=1200 + P + ql)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    o_ref, _, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1],
                                     pos, orc.causal_mask(B, ql, N, dt), dt)
    out, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False)
    np.testing.assert_allclose(out, o_ref, **OUT_TOL[dt])
