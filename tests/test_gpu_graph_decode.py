"""The graph-capturable decode step (ABI 3): device-resident step state, the decode kernel's device-length form against
its static form (bit for bit) and the oracle, and DecodeGraph — one captured HIP graph of the whole patched layer stack
replayed per token — against the eager per-token loop of the plugin (run_spatten_llama.py:27-35).  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, TORCH_DT, attn_inputs, check_stash, dev, host

pytestmark = pytest.mark.gpu


def test_step_state_set_and_advance():
    from spatten_amd import ops
    for dt, d in (("bf16", 128), ("f16", 64), ("f32", 128), ("bf16", 256)):
        cos, sin = ops.rope_table(300, d, TORCH_DT[dt], "cuda")
        st = ops.StepState(cos, sin)
        st.set(17, 16)
        assert st.read() == (17, 16)
        h = d // 2
        es = cos.element_size()
        rows = lambda: st.buf[64:64 + 4 * h * es].view(TORCH_DT[dt]).view(4, h)
        assert torch.equal(rows()[0], cos[16]) and torch.equal(rows()[1], cos[16])
        assert torch.equal(rows()[2], sin[16]) and torch.equal(rows()[3], sin[16])
        st.advance()
        st.advance(2)
        assert st.read() == (20, 19)
        assert torch.equal(rows()[0], cos[19]) and torch.equal(rows()[1], cos[19]) and torch.equal(rows()[3], sin[19])
        st.set(40, 7)                      # query position and appended slot differ: two distinct staged rows
        assert torch.equal(rows()[0], cos[7]) and torch.equal(rows()[1], cos[39])
        with pytest.raises(ValueError):
            st.set(301, 300)


@pytest.mark.parametrize("dt,d,H,Hkv,P,cap", [
    ("bf16", 128, 32, 32, 2050, 2176),      # the C2 turn: lean kernel, 8 splits laid out for the capacity
    ("bf16", 128, 4, 4, 700, 1024),         # few heads: 64 splits, most of them short; later ones EMPTY at this length
    ("f16", 64, 8, 2, 333, 512),            # GQA: the general kernel
    ("f32", 128, 2, 2, 95, 128),
    ("bf16", 128, 8, 8, 5000, 5120),        # long chunks: the double-buffered instantiation
    ("bf16", 256, 2, 2, 130, 256),
])
def test_device_length_steps_equal_static_steps_bitwise(dt, d, H, Hkv, P, cap):
    """Six consecutive tokens: the device-length form (one argument block, the state advanced on the device) against the
    static form laid out for the same length (kv_len_layout) — outputs, stash, appended rows bit for bit — and the first
    step against the oracle."""
    from spatten_amd import ops
    B, tdt, steps = 1, TORCH_DT[dt], 6
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, 1, dt, 11)
    cos_h, sin_h = orc.rope_table(cap + 8, d, dt)
    cos, sin = dev(cos_h[:, : d // 2], dt), dev(sin_h[:, : d // 2], dt)

    def planes():
        kc = torch.zeros(B, Hkv, cap, d, dtype=tdt, device="cuda")
        vc, krc = torch.zeros_like(kc), torch.zeros_like(kc)
        kc[:, :, :P], vc[:, :, :P] = dev(past[0], dt), dev(past[1], dt)
        ops.build_shadow(kc, krc, 0, P, cos, sin)
        # stale but FINITE rows past the length (an older, longer turn): must not leak into the result
        krc[:, :, P + steps:] = 3.0
        vc[:, :, P + steps:] = -2.0
        return kc, krc, vc
    ka, kra, va = planes()
    kb, krb, vb = planes()
    st = ops.StepState(cos, sin)
    st.set(P, P - 1)
    stash_a = torch.zeros(B, H, cap, dtype=tdt, device="cuda")
    stash_b = torch.zeros_like(stash_a)
    qs = [dev(orc.synth_normal(50 + t, 0, (B, H, d), dt), dt) for t in range(steps)]
    ks = [dev(orc.synth_normal(50 + t, 1, (B, Hkv, d), dt), dt) for t in range(steps)]
    vs = [dev(orc.synth_normal(50 + t, 2, (B, Hkv, d), dt), dt) for t in range(steps)]
    qs[0], ks[0], vs[0] = dev(q[:, :, 0], dt), dev(k[:, :, 0], dt), dev(v[:, :, 0], dt)
    for t in range(steps):
        n = P + t + 1
        st.advance()
        oa = ops.attn_decode(qs[t], ka, kra, va, cap, cos, sin, 0, k_new=ks[t], v_new=vs[t], scores=stash_a, step=st)
        ob = ops.attn_decode(qs[t], kb, krb, vb, n, cos, sin, n - 1, k_new=ks[t], v_new=vs[t], scores=stash_b, layout=cap)
        torch.cuda.synchronize()
        assert st.read() == (n, n - 1)
        assert torch.equal(oa, ob), (t, (oa.float() - ob.float()).abs().max().item())
        assert torch.equal(stash_a[:, :, :n], stash_b[:, :, :n]), t
        assert (stash_a[:, :, n:] == 0).all(), "stash written past the length"
        assert torch.equal(ka[:, :, :n], kb[:, :, :n]) and torch.equal(kra[:, :, :n], krb[:, :, :n]) and torch.equal(va[:, :, :n], vb[:, :, :n])
        if t == 0:
            o, stash, _ = orc.attention_core(q, k, v, past[0], past[1], np.full((B, 1), P), None, dt)
            np.testing.assert_allclose(host(oa)[:, None], o, **OUT_TOL[dt])
            check_stash(host(stash_a[:, :, :n])[:, :, None], stash, dt, "device-length step")
    assert (kra[:, :, P + steps:] == 3.0).all() and (va[:, :, P + steps:] == -2.0).all(), "rows past the length were written"


def test_device_length_step_rejects_what_it_does_not_cover():
    from spatten_amd import ops
    cos, sin = ops.rope_table(256, 128, torch.bfloat16, "cuda")
    st = ops.StepState(cos, sin)
    st.set(10, 9)
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")
    q, kc = z(1, 4, 128), z(1, 4, 128, 128)
    with pytest.raises(RuntimeError):       # a mask needs a host length: the argument check refuses the combination
        ops.attn_decode(q, kc, kc.clone(), kc.clone(), 128, cos, sin, 0, k_new=z(1, 4, 128), v_new=z(1, 4, 128),
                        mask=z(1, 128), scores=z(1, 4, 128), step=st)
    with pytest.raises(ValueError):         # the stash row must cover the bound
        ops.attn_decode(q, kc, kc.clone(), kc.clone(), 128, cos, sin, 0, k_new=z(1, 4, 128), v_new=z(1, 4, 128),
                        scores=z(1, 4, 64), step=st)


@pytest.mark.parametrize("units,kv_len", [(4, 20665), (5, 16401), (6, 13449)])
def test_appended_token_alone_in_the_last_split_of_the_pipelined_kernel(units, kv_len):
    """Round-2 advisor finding: with a chunk above one single-shot tile and N = (S-1)*chunk + 1 the appended token is the
    only row of the last split; the double-buffered loop ran zero times and the token was never scored or stored."""
    from spatten_amd import _lib, ops
    dt, tdt, d, B, H = "bf16", torch.bfloat16, 128, 1, units
    S = _lib.load().spatten_decode_auto_splits(B, H, d, kv_len)
    per = -(-kv_len // S)
    chunk = -(-per // 8) * 8                # decode_rows: balanced chunks, rounded to the 8 rows of a stash line
    assert (kv_len - 1) % chunk == 0 and chunk > 320, "shape no longer hits the empty owning split"
    P = kv_len - 1
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(tdt)
    cos, sin = ops.rope_table(kv_len + 8, d, tdt, "cuda")
    kc = torch.full((B, H, kv_len + 3, d), float("nan"), dtype=tdt, device="cuda")
    vc, krc = kc.clone(), kc.clone()
    kc[:, :, :P], vc[:, :, :P] = rnd(B, H, P, d), rnd(B, H, P, d)
    ops.build_shadow(kc, krc, 0, P, cos, sin)
    q, kn, vn = rnd(B, H, d) * 2, rnd(B, H, d) * 2, rnd(B, H, d)
    outs, stashes = [], []
    for ns in (0, 1):                       # the auto split count (hits the case) against a single workgroup per head
        k2, kr2, v2 = kc.clone(), krc.clone(), vc.clone()
        stash = torch.full((B, H, kv_len), float("nan"), dtype=tdt, device="cuda")
        o = ops.attn_decode(q, k2, kr2, v2, kv_len, cos, sin, P, k_new=kn, v_new=vn, scores=stash, n_splits=ns)
        torch.cuda.synchronize()
        assert not torch.isnan(stash.float()).any(), "the appended token's logit is missing from the stash"
        assert torch.equal(k2[:, :, P], kn) and torch.equal(v2[:, :, P], vn), "appended rows not stored"
        assert torch.equal(kr2[:, :, P:P + 1], ops.rope_single(kn[:, :, None], cos, sin, pos0=P))
        outs.append(o)
        stashes.append(stash)
    assert torch.equal(stashes[0], stashes[1])
    np.testing.assert_allclose(host(outs[0]), host(outs[1]), **OUT_TOL[dt])


def test_pv_gather_many_units_with_a_workspace():
    """Round-2 advisor finding: k * B * H above ~1M made the split P.V gather return UNSUPPORTED (B = 8, H = 32, k = 4915)."""
    from spatten_amd import ops
    B, H, d, N, k = 8, 32, 128, 8192, 4915
    tdt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    stash = torch.randn(B, H, N, device="cuda", generator=g).to(tdt)
    V = torch.randn(B, H, N, d, device="cuda", generator=g).to(tdt)
    lse = torch.stack([stash.float().amax(-1), torch.exp(stash.float() - stash.float().amax(-1, keepdim=True)).sum(-1)], -1).contiguous()
    idx = torch.stack([torch.randperm(N, device="cuda", generator=g)[:k].sort().values for _ in range(B * H)]).to(torch.int32)
    out = ops.pv_gather(stash, lse, V, idx)
    torch.cuda.synchronize()
    p = torch.softmax(stash.float(), -1)
    ix = idx.view(B, H, k).long()
    want = torch.einsum("bhk,bhkd->bhd", p.gather(2, ix), V.float().gather(2, ix[..., None].expand(B, H, k, d)))
    np.testing.assert_allclose(host(out).reshape(B, H, d), host(want), atol=2e-3, rtol=2e-2)


# ------------------------------------------------------------------------------------------------------------------
# DecodeGraph on the plugin surface
# ------------------------------------------------------------------------------------------------------------------
LAYERS, H, D = 3, 8, 128
HID = H * D


class LlamaAttention(nn.Module):          # duck-typed by class name, like HF's module
    def __init__(self, dt):
        super().__init__()
        self.config = SimpleNamespace(pretraining_tp=1)
        self.num_heads = self.num_key_value_heads = H
        self.num_key_value_groups, self.head_dim, self.hidden_size = 1, D, HID
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self, n, nn.Linear(HID, HID, bias=False, dtype=dt, device="cuda"))


class Stack(nn.Module):
    """The attention path of a decoder stack under the transformers 4.33 calling convention (device-built mask and
    position_ids, legacy (K, V) tuples)."""

    def __init__(self, dt):
        super().__init__()
        self.config = SimpleNamespace(model_type="llama")
        self.layers = nn.ModuleList([LlamaAttention(dt) for _ in range(LAYERS)])

    @torch.no_grad()
    def forward(self, x, past):
        B, q, _ = x.shape
        P = 0 if past is None else past[0][0].shape[2]
        N = P + q
        pos = torch.arange(P, N, device=x.device)[None]
        mask = torch.zeros(B, 1, q, N, dtype=x.dtype, device=x.device)
        if q > 1:
            mask.masked_fill_(torch.ones(q, N, dtype=torch.bool, device=x.device).triu(P + 1), torch.finfo(x.dtype).min)
        new_past = []
        for i, m in enumerate(self.layers):
            a, _, kv = m(x, attention_mask=mask, position_ids=pos, past_key_value=None if past is None else past[i], use_cache=True)
            x = x + a
            new_past.append(kv)
        return x, new_past


def _models(dt, **kw):
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    torch.manual_seed(0)
    a = Stack(dt)
    for p in a.parameters():
        p.data.mul_(0.5)
    b = Stack(dt)
    b.load_state_dict(a.state_dict())
    caches = []
    for m in (a, b):
        with contextlib.redirect_stdout(io.StringIO()):
            caches.append(enable_spatten_llm(m, 4, 60, 64, **kw))
    return a, b, caches


@pytest.mark.parametrize("dt,kw", [(torch.bfloat16, {}), (torch.float32, {}),
                                   (torch.bfloat16, dict(native_gemv=True)), (torch.float16, dict(native_gemv=True, fuse_qkv=True))])
def test_decode_graph_replays_equal_the_eager_plugin_loop_bitwise(dt, kw):
    """The same tokens through (a) the eager per-token loop of the patched forward and (b) DecodeGraph: one eager warm-up
    step, one capture, then replays of ONE graph of the whole layer stack.  Hidden states of every token, the final
    K / V caches and every module's attn_scores must agree bit for bit; then a prune event on both (bit exact) and a second
    turn on a NEW graph."""
    from spatten_amd import kv_slab
    from spatten_amd.graph import DecodeGraph
    a, b, (cache_a, cache_b) = _models(dt, **kw)
    g = torch.Generator(device="cuda").manual_seed(1)
    P, T = 200, 14
    x0 = torch.randn(1, P, HID, device="cuda", generator=g).to(dt)
    toks = [torch.randn(1, 1, HID, device="cuda", generator=g).to(dt) for _ in range(2 * T + 8)]
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=T)
        for t in range(T):
            x = toks[turn * T + t]
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), (turn, t, (ya.float() - yb.float()).abs().max().item())
        assert graph.n_replays == T - 1, "every step after the warm-up must be a replay of the one captured graph"
        past_b = graph.past_key_values
        n = past_a[0][0].shape[2]
        assert n == P + (turn + 1) * T - (0 if turn == 0 else pruned)
        for (ka, va), (kb, vb), ma, mb in zip(past_a, past_b, a.layers, b.layers):
            assert kb.shape == ka.shape and torch.equal(ka, kb) and torch.equal(va, vb)
            sa, sb = kv_slab.slab_of(ka), kv_slab.slab_of(kb)
            assert torch.equal(sa.kr[:, :, :n], sb.kr[:, :, :n])
            assert mb.attn_scores.shape == ma.attn_scores.shape and torch.equal(ma.attn_scores, mb.attn_scores)
        if turn == 0:       # the reference's turn boundary (run_spatten_llama.py:71-79): prune from the last step's stash
            coming = 8 + T
            new_a = cache_a.apply_token_pruning(past_a, coming, [m.attn_scores for m in a.layers])
            new_b = cache_b.apply_token_pruning(past_b, coming, [m.attn_scores for m in b.layers])
            assert new_a is not past_a
            pruned = n - new_a[0][0].shape[2]
            for (ka, va), (kb, vb) in zip(new_a, new_b):
                assert torch.equal(ka, kb) and torch.equal(va, vb)
            xp = torch.randn(1, 8, HID, device="cuda", generator=g).to(dt)      # the next turn's prompt
            _, past_a = a(xp, new_a)
            _, past_b = b(xp, new_b)
            pruned -= 8


def test_decode_graph_outgrows_its_slabs_and_recaptures():
    from spatten_amd.graph import DecodeGraph
    dt = torch.bfloat16
    a, b, _ = _models(dt)
    g = torch.Generator(device="cuda").manual_seed(2)
    x0 = torch.randn(1, 100, HID, device="cuda", generator=g).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=8)
    bound0 = graph.bound
    T = bound0 - 100 + 5                    # runs past the capacity the slabs had at capture
    for t in range(T):
        x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
        ya, past_a = a(x, past_a)
        yb = graph.step(x)
        # slabs of different capacity split the keys differently: same result up to the merge order
        np.testing.assert_allclose(host(yb), host(ya), atol=2e-2, rtol=2e-2)
    assert graph.bound > bound0 and graph.length == 100 + T
    for (ka, va), (kb, vb) in zip(past_a, graph.past_key_values):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


def test_native_gemv_forward_stays_within_rounding_of_the_torch_projections():
    """enable_spatten_llm(native_gemv=True): the single-token projections on the streaming kernel — same fp32
    accumulation and single rounding as nn.Linear, another summation order."""
    dt = torch.bfloat16
    a, b, _ = _models(dt)
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    with contextlib.redirect_stdout(io.StringIO()):
        enable_spatten_llm(b, 4, 60, 64, native_gemv=True, fuse_qkv=True)
    g = torch.Generator(device="cuda").manual_seed(4)
    x0 = torch.randn(1, 150, HID, device="cuda", generator=g).to(dt)
    ya, past_a = a(x0, None)
    yb, past_b = b(x0, None)
    assert torch.equal(ya, yb), "multi-token forwards keep torch's GEMMs"
    for t in range(5):
        x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
        ya, past_a = a(x, past_a)
        yb, past_b = b(x, past_b)
        np.testing.assert_allclose(host(yb), host(ya), atol=6e-2, rtol=3e-2)
    # model.half(): the stacked q/k/v weight is rebuilt from the new parameters instead of going stale
    b.half()
    yh, _ = b(x0.half(), None)
    yh1, _ = b(x.half(), _)
    assert yh.dtype == torch.float16 and torch.isfinite(yh1.float()).all()


def test_decode_graph_holds_the_head_parallel_exchange():
    """Head-parallel plugin + DecodeGraph: with the library-owned RCCL communicator initialised (world size 1 here — a
    multi-GPU node is the driver's) the all-gather in front of o_proj is an ordinary stream operation INSIDE the captured
    decode step; replays equal the eager loop bit for bit."""
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    from spatten_amd.graph import DecodeGraph
    from spatten_amd.parallel import HeadParallel
    dt = torch.bfloat16
    a, b, _ = _models(dt)
    hp = HeadParallel(H)
    try:
        hp.init_native()
    except RuntimeError as e:
        assert "unsupported" in str(e)
        pytest.skip("librccl not installed")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            enable_spatten_llm(b, 4, 60, 64, head_parallel=hp)
        g = torch.Generator(device="cuda").manual_seed(6)
        x0 = torch.randn(1, 90, HID, device="cuda", generator=g).to(dt)
        _, past_a = a(x0, None)
        _, past_b = b(x0, None)
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=10)
        for t in range(8):
            x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), t
        assert graph.n_replays == 7
    finally:
        hp.close_native()


@pytest.mark.parametrize("kw", [dict(importance_mode="cascade"), dict(head_keep=[6, 5, 5]),
                                dict(importance_mode="cascade", head_keep=6, fuse_qkv=True, native_gemv=True),
                                dict(pq_threshold=0.05), dict(pq_threshold=0.02, head_keep=6, fuse_qkv=True),
                                dict(pq_threshold=0.05, importance_mode="cascade"),
                                dict(pq_threshold=0.05, pq_profile=(4, 8)), dict(pq_threshold=0.02, pq_profile=(8, 8), head_keep=6),
                                dict(pq_threshold=0.05, pq_profile=(6, 6), fuse_qkv=True),
                                dict(local_v_keep=0.4), dict(local_v_keep=0.3, head_keep=6),
                                dict(layer_keep=[36, 30, 24]), dict(layer_keep=[36, 30, 30], importance_mode="cascade", head_keep=7)])
def test_decode_graph_covers_cascade_importance_and_head_pruning(kw):
    """The SpAtten modes whose decode step is ONE fused launch — cumulative importance (the previous step's probabilities
    folded while this step streams; under the graph the two stash buffers swap roles on the device) and head pruning
    (head list + fused head importance) — replayed from one captured graph: hidden states of every token equal the eager
    loop's bit for bit, and so do the accumulators, the head scores and the prune event that follows (kept positions,
    kept heads, compacted caches); then a second turn on the pruned caches."""
    from spatten_amd.graph import DecodeGraph
    dt = torch.bfloat16
    a, b, (cache_a, cache_b) = _models(dt, **kw)
    g = torch.Generator(device="cuda").manual_seed(11)
    P, T = 180, 9
    x0 = torch.randn(1, P, HID, device="cuda", generator=g).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=T)
        for t in range(T):
            x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), (turn, t, (ya.float() - yb.float()).abs().max().item())
        assert graph.n_replays == T - 1
        past_b = graph.past_key_values
        n = past_a[0][0].shape[2]
        for ma, mb, la in zip(a.layers, b.layers, cache_a.ext.layers):
            if la.head_ids is None:
                assert torch.equal(ma.attn_scores, mb.attn_scores)
            else:           # pruned heads are not launched: their stash rows are whatever an earlier step left there
                kept = la.head_ids.long()
                assert torch.equal(ma.attn_scores[:, kept], mb.attn_scores[:, kept])
        if "pq_threshold" in kw:    # progressive quantisation: the planes every replayed step packed, and the refetch flags
            from spatten_amd import kv_slab
            flagged = 0
            for (ka, _), (kb, _), la, lb in zip(past_a, past_b, cache_a.ext.layers, cache_b.ext.layers):
                sa, sb = kv_slab.slab_of(ka), kv_slab.slab_of(kb)
                assert sa.pq_len == n and sb.pq_len == n
                assert torch.equal(sa.pq.msb[:, :, :n], sb.pq.msb[:, :, :n]) and torch.equal(sa.pq.lsb[:, :, :n], sb.pq.lsb[:, :, :n])
                assert torch.equal(sa.pq.scale[:, :, :n], sb.pq.scale[:, :, :n])
                if "pq_profile" in kw:      # the quantised value plane of the profile
                    assert (sa.pq.key_bits, sa.pq.value_bits) == tuple(kw["pq_profile"])
                    assert torch.equal(sa.pq.vq[:, :, :n], sb.pq.vq[:, :, :n]) and torch.equal(sa.pq.vscale[:, :, :n], sb.pq.vscale[:, :, :n])
                heads = slice(None) if la.head_ids is None else la.head_ids.long()
                assert torch.equal(la.need_lsb[heads], lb.need_lsb[heads])
                flagged += int(la.need_lsb[heads].sum())
            assert flagged > 0          # the threshold is chosen so that both the confident and the refetch branch run
        coming = 6 + T
        launched = [slice(None) if la.head_ids is None else la.head_ids.long() for la in cache_a.ext.layers]   # heads of this turn
        new_a = cache_a.apply_token_pruning(past_a, coming, [m.attn_scores for m in a.layers])
        new_b = cache_b.apply_token_pruning(past_b, coming, [m.attn_scores for m in b.layers])
        assert (new_a is not past_a) == (new_b is not past_b) and (turn > 0 or new_a is not past_a)
        for la, lb, hk in zip(cache_a.ext.layers, cache_b.ext.layers, launched):
            if cache_a.ext.cascade:
                assert torch.equal(la.acc[hk, :new_a[0][0].shape[2]], lb.acc[hk, :new_a[0][0].shape[2]])
            assert torch.equal(la.head_abs, lb.head_abs)
            assert (la.head_ids is None and lb.head_ids is None) or torch.equal(la.head_ids, lb.head_ids)
        for (ka, va), (kb, vb), la in zip(new_a, new_b, cache_a.ext.layers):
            kept = slice(None) if la.head_ids is None else la.head_ids.long()
            assert torch.equal(ka[:, kept], kb[:, kept]) and torch.equal(va[:, kept], vb[:, kept])
        if "head_keep" in kw:
            assert any(st.head_ids is not None for st in cache_b.ext.layers)
        xp = torch.randn(1, 6, HID, device="cuda", generator=g).to(dt)
        _, past_a = a(xp, new_a)
        _, past_b = b(xp, new_b)


class _RMSNorm(nn.Module):
    def __init__(self, n, dt):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n, dtype=dt, device="cuda"))

    def forward(self, x):
        v = x.float()
        return (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)).to(x.dtype) * self.weight


class _DecoderLayer(nn.Module):
    def __init__(self, dt):
        super().__init__()
        self.self_attn = LlamaAttention(dt)
        self.ln1, self.ln2 = _RMSNorm(HID, dt), _RMSNorm(HID, dt)
        self.up = nn.Linear(HID, 2 * HID, bias=False, dtype=dt, device="cuda")
        self.down = nn.Linear(2 * HID, HID, bias=False, dtype=dt, device="cuda")


class _TinyLM(nn.Module):
    """A whole (small) Llama-like decoder — embedding, pre-norm blocks with the patched attention and an MLP, final norm,
    lm_head — under the transformers 4.33 calling convention (device-built mask / position_ids)."""
    VOCAB = 211

    def __init__(self, dt):
        super().__init__()
        self.config = SimpleNamespace(model_type="llama")
        self.embed = nn.Embedding(self.VOCAB, HID, dtype=dt, device="cuda")
        self.layers = nn.ModuleList([_DecoderLayer(dt) for _ in range(LAYERS)])
        self.norm = _RMSNorm(HID, dt)
        self.lm_head = nn.Linear(HID, self.VOCAB, bias=False, dtype=dt, device="cuda")

    @torch.no_grad()
    def forward(self, ids, past):
        B, q = ids.shape
        P = 0 if past is None else past[0][0].shape[2]
        pos = torch.arange(P, P + q, device=ids.device)[None]
        mask = torch.zeros(B, 1, q, P + q, dtype=self.embed.weight.dtype, device=ids.device)
        if q > 1:
            mask.masked_fill_(torch.ones(q, P + q, dtype=torch.bool, device=ids.device).triu(P + 1), torch.finfo(mask.dtype).min)
        x = self.embed(ids)
        new_past = []
        for i, layer in enumerate(self.layers):
            a, _, kv = layer.self_attn(layer.ln1(x), attention_mask=mask, position_ids=pos,
                                       past_key_value=None if past is None else past[i], use_cache=True)
            x = x + a
            x = x + layer.down(torch.nn.functional.silu(layer.up(layer.ln2(x))))
            new_past.append(kv)
        return self.lm_head(self.norm(x)), new_past


def test_greedy_decoding_of_a_whole_model_under_one_graph():
    """The reference's greedy_generate (run_spatten_llama.py:18-57) with the per-token `model(...)` call replaced by
    DecodeGraph.step: embedding, norms, MLPs, the patched attention of every layer and the lm_head are ONE captured graph;
    the argmax token is fed back on the device.  Same tokens, same logits (bit for bit) as the eager loop; then the prune of
    the turn boundary from the graph's stashes."""
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    from spatten_amd.graph import DecodeGraph
    dt = torch.bfloat16
    torch.manual_seed(3)
    a = _TinyLM(dt)
    b = _TinyLM(dt)
    b.load_state_dict(a.state_dict())
    caches = []
    for m in (a, b):
        with contextlib.redirect_stdout(io.StringIO()):
            caches.append(enable_spatten_llm(m, 4, 40, 40, native_gemv=True))
    prompt = torch.randint(0, _TinyLM.VOCAB, (1, 120), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    la, past_a = a(prompt, None)
    lb, past_b = b(prompt, None)
    assert torch.equal(la, lb)
    tok_a = la[:, -1:].argmax(-1)
    tok_b = lb[:, -1:].argmax(-1)
    graph = DecodeGraph(lambda past, ids: tuple(reversed(b(ids, past))), past_b, horizon=24)
    toks_a, toks_b = [], []
    for t in range(20):
        la, past_a = a(tok_a, past_a)
        lb = graph.step(tok_b)
        assert torch.equal(la, lb), t
        tok_a, tok_b = la[:, -1:].argmax(-1), lb[:, -1:].argmax(-1)       # stays on the device: no host sync per token
        toks_a.append(tok_a)
        toks_b.append(tok_b.clone())
    assert torch.equal(torch.cat(toks_a, 1), torch.cat(toks_b, 1)) and graph.n_replays == 19
    past_b = graph.past_key_values
    new_a = caches[0].apply_token_pruning(past_a, 30, [l.self_attn.attn_scores for l in a.layers])
    new_b = caches[1].apply_token_pruning(past_b, 30, [l.self_attn.attn_scores for l in b.layers])
    for (ka, va), (kb, vb) in zip(new_a, new_b):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


class _LMOutput:
    def __init__(self, logits=None, past_key_values=None):
        self.logits, self.past_key_values = logits, past_key_values


class _HFStyleLM(_TinyLM):
    """_TinyLM under the keyword calling convention of a transformers causal LM (what run_spatten_llama.py calls)."""

    def forward(self, input_ids=None, past_key_values=None, use_cache=None, attention_mask=None, position_ids=None):
        logits, new_past = _TinyLM.forward(self, input_ids, past_key_values)
        return _LMOutput(logits=logits, past_key_values=new_past)


@pytest.mark.parametrize("kw", [dict(), dict(importance_mode="cascade", head_keep=6)])
def test_the_reference_loop_unchanged_replays_one_graph_per_token(kw):
    """enable_spatten_llm(..., auto_graph=True): the reference's caller (run_spatten_llama.py:18-57 greedy_generate, :60-87
    the turn protocol — collect m.attn_scores, apply_token_pruning, prefill the next prompt, decode token by token with
    keyword calls and a host read of every token) runs UNCHANGED and every single-token call after the first two of a turn
    is a graph replay.  Tokens, logits, stashes and pruned caches equal the same loop without auto_graph, bit for bit."""
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    dt = torch.bfloat16
    torch.manual_seed(5)
    a, b = _HFStyleLM(dt), _HFStyleLM(dt)
    b.load_state_dict(a.state_dict())
    caches = []
    for m, auto in ((a, False), (b, True)):
        with contextlib.redirect_stdout(io.StringIO()):
            caches.append(enable_spatten_llm(m, 4, 40, 40, auto_graph=auto, **kw))
    assert hasattr(b, "_spatten_auto_graph") and not hasattr(a, "_spatten_auto_graph")

    def greedy_generate(model, input_ids, past_key_values, max_gen_len):          # run_spatten_llama.py:18-57
        outputs = model(input_ids=input_ids, past_key_values=past_key_values, use_cache=True)
        past_key_values = outputs.past_key_values
        pred = outputs.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
        generated, logits = [pred.item()], [outputs.logits[:, -1].clone()]
        for _ in range(max_gen_len - 1):
            outputs = model(input_ids=pred, past_key_values=past_key_values, use_cache=True)
            past_key_values = outputs.past_key_values
            pred = outputs.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
            generated.append(pred.item())
            logits.append(outputs.logits[:, -1].clone())
        return past_key_values, generated, logits

    def inference(model, kv_cache, prompts, max_gen_len):                         # run_spatten_llama.py:60-87
        past, trace = None, []
        for idx, ids in enumerate(prompts):
            if idx > 0:
                scores = [m.self_attn.attn_scores for m in model.layers]
                n_prev = past[0][0].size(2)
                past = kv_cache.apply_token_pruning(past, ids.shape[1] + max_gen_len, scores)
                trace.append((n_prev, past[0][0].size(2)))
            past, gen, logits = greedy_generate(model, ids, past, max_gen_len)
            trace.append((gen, logits))
        return past, trace

    g = torch.Generator(device="cuda").manual_seed(21)
    prompts = [torch.randint(0, _TinyLM.VOCAB, (1, n), device="cuda", generator=g) for n in (90, 17, 23)]
    with torch.no_grad():
        past_a, tr_a = inference(a, caches[0], prompts, 12)
        past_b, tr_b = inference(b, caches[1], prompts, 12)
    for ea, eb in zip(tr_a, tr_b):
        if isinstance(ea[0], int):
            assert ea == eb                                  # cache lengths before / after each prune
        else:
            assert ea[0] == eb[0]                            # the generated tokens
            assert all(torch.equal(x, y) for x, y in zip(ea[1], eb[1]))
    ext = getattr(caches[0], "ext", None)      # (pruned heads are not launched: their rows are whatever an earlier step left)
    kept = [slice(None) if ext is None or st.head_ids is None else st.head_ids.long() for st in (ext.layers if ext else a.layers)]
    for (ka, va), (kb, vb), hk in zip(past_a, past_b, kept):
        assert torch.equal(ka[:, hk], kb[:, hk]) and torch.equal(va[:, hk], vb[:, hk])
    for la, lb, hk in zip(a.layers, b.layers, kept):
        assert torch.equal(la.self_attn.attn_scores[:, hk], lb.self_attn.attn_scores[:, hk])
    graph = b._spatten_auto_graph["graph"]
    assert graph is not None and graph.n_replays == 12 - 1 - 1       # 11 single-token calls per turn: 1 eager, then capture + replays


@pytest.mark.parametrize("dt,d", [(torch.bfloat16, 128), (torch.float16, 64), (torch.float32, 128)])
def test_kv_append_step_equals_append_plus_pack_bitwise(dt, d):
    """spatten_kv_append_step (row from the device state, rotary row staged in the state) leaves in k / kr / v and in the
    progressive-quant planes exactly what spatten_kv_append + spatten_pq_pack leave with host lengths; a row at or past the
    capacity is not written."""
    from spatten_amd import ops
    B, Hkv, cap = 2, 3, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(dt)
    cos, sin = ops.rope_table(cap + 8, d, dt, "cuda")
    ka, kra, va = rnd(B, Hkv, cap, d), rnd(B, Hkv, cap, d), rnd(B, Hkv, cap, d)
    kb, krb, vb = ka.clone(), kra.clone(), va.clone()
    pa, pb = ops.PQPlanes(B, Hkv, cap, d, "cuda"), ops.PQPlanes(B, Hkv, cap, d, "cuda")
    st = ops.StepState(cos, sin)
    st.set(10, 9)
    for n in range(11, 15):
        kn, vn = rnd(B, Hkv, d), rnd(B, Hkv, d)
        ops.kv_append(kn[:, :, None], vn[:, :, None], ka, kra, va, n - 1, cos, sin)
        ops.pq_pack(kra, pa, n - 1, n)
        st.advance(1)
        ops.kv_append_step(kn, vn, kb, krb, vb, st, pb)
    torch.cuda.synchronize()
    assert torch.equal(ka, kb) and torch.equal(kra, krb) and torch.equal(va, vb)
    assert torch.equal(pa.msb, pb.msb) and torch.equal(pa.lsb, pb.lsb) and torch.equal(pa.scale, pb.scale)
    st.set(cap + 1, cap)                    # row = capacity: nothing may be written
    before = (kb.clone(), krb.clone(), vb.clone(), pb.msb.clone())
    ops.kv_append_step(rnd(B, Hkv, d), rnd(B, Hkv, d), kb, krb, vb, st, pb)
    ops.kv_append_step(rnd(B, Hkv, d), rnd(B, Hkv, d), None, krb, vb, st, None)     # without planes / un-rotated plane
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(before, (kb, krb, vb, pb.msb)))


def test_decode_graph_at_llama2_7b_geometry_equals_the_eager_loop_bitwise(monkeypatch):
    """The graph path at BASELINE.json configs[1] scale — 32 heads x 128, a 4096-token cache pruned to 2048 (start 4 /
    important 1020 / recent 1024), 64-token turn — on a 2-layer stack: prune event from the stashes of the turn, prompt
    prefill, then every decode step of the next turn through DecodeGraph (stacked q/k/v + native projections) against the
    eager loop: hidden states, stashes and caches bit for bit."""
    import contextlib
    import io
    import sys

    from spatten_amd import enable_spatten_llm
    from spatten_amd.graph import DecodeGraph
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "LAYERS", 2)
    monkeypatch.setattr(mod, "H", 32)
    monkeypatch.setattr(mod, "HID", 32 * 128)
    dt = torch.bfloat16
    torch.manual_seed(1)
    a = Stack(dt)
    for p in a.parameters():
        p.data.mul_(0.25)
    b = Stack(dt)
    b.load_state_dict(a.state_dict())
    caches = []
    for m in (a, b):
        with contextlib.redirect_stdout(io.StringIO()):
            caches.append(enable_spatten_llm(m, 4, 1020, 1024, fuse_qkv=True, native_gemv=True, assume_causal=True))
    g = torch.Generator(device="cuda").manual_seed(2)
    x0 = (torch.randn(1, 4096 - 64, 32 * 128, device="cuda", generator=g) * 0.5).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=64)
        for t in range(64 if turn == 0 else 8):
            x = (torch.randn(1, 1, 32 * 128, device="cuda", generator=g) * 0.5).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            assert torch.equal(ya, yb), (turn, t)
        past_b = graph.past_key_values
        for la, lb in zip(a.layers, b.layers):
            assert torch.equal(la.attn_scores, lb.attn_scores)
        for (ka, va), (kb, vb) in zip(past_a, past_b):
            assert torch.equal(ka, kb) and torch.equal(va, vb)
        if turn == 0:
            assert past_a[0][0].shape[2] == 4096
            new_a = caches[0].apply_token_pruning(past_a, 128, [m.attn_scores for m in a.layers])
            new_b = caches[1].apply_token_pruning(past_b, 128, [m.attn_scores for m in b.layers])
            assert new_a[0][0].shape[2] == new_b[0][0].shape[2] <= 2048          # 4 + 1020 + what is left of the recent window
            for (ka, va), (kb, vb) in zip(new_a, new_b):
                assert torch.equal(ka, kb) and torch.equal(va, vb)
            xp = (torch.randn(1, 64, 32 * 128, device="cuda", generator=g) * 0.5).to(dt)
            _, past_a = a(xp, new_a)
            _, past_b = b(xp, new_b)


@pytest.mark.parametrize("kw", [dict(), dict(fuse_qkv=True, native_gemv=True), dict(importance_mode="cascade"),
                                dict(head_keep=6), dict(pq_threshold=0.05), dict(layer_keep=[50, 40, 30]),
                                dict(importance_mode="cascade", head_keep=6, pq_threshold=0.05)])
def test_decode_graph_on_a_grouped_query_stack(kw):
    """Grouped-query attention (num_key_value_heads < num_heads: k_proj / v_proj are narrower, the cache holds Hkv heads,
    modify_llama.py:106-108 repeat_kv) through DecodeGraph against the eager loop, with a prune event on the GQA cache."""
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    from spatten_amd.graph import DecodeGraph
    dt, Hkv = torch.bfloat16, 2
    torch.manual_seed(7)

    def gqa_stack():
        st = Stack(dt)
        for m in st.layers:
            m.num_key_value_heads, m.num_key_value_groups = Hkv, H // Hkv
            m.k_proj = nn.Linear(HID, Hkv * D, bias=False, dtype=dt, device="cuda")
            m.v_proj = nn.Linear(HID, Hkv * D, bias=False, dtype=dt, device="cuda")
        return st
    a = gqa_stack()
    for p in a.parameters():
        p.data.mul_(0.5)
    b = gqa_stack()
    b.load_state_dict(a.state_dict())
    caches = []
    for m in (a, b):
        with contextlib.redirect_stdout(io.StringIO()):
            caches.append(enable_spatten_llm(m, 4, 60, 64, **kw))
    g = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(1, 170, HID, device="cuda", generator=g).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    assert past_a[0][0].shape[1] == Hkv
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=8)
        for t in range(8):
            x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            assert torch.equal(ya, yb), (turn, t)
        past_b = graph.past_key_values
        new_a = caches[0].apply_token_pruning(past_a, 14, [m.attn_scores for m in a.layers])
        new_b = caches[1].apply_token_pruning(past_b, 14, [m.attn_scores for m in b.layers])
        assert new_a is not past_a
        for (ka, va), (kb, vb) in zip(new_a, new_b):
            assert ka.shape[1] == Hkv and torch.equal(ka, kb) and torch.equal(va, vb)
        xp = torch.randn(1, 6, HID, device="cuda", generator=g).to(dt)
        ya, past_a = a(xp, new_a)
        yb, past_b = b(xp, new_b)
        assert torch.equal(ya, yb)


@pytest.mark.parametrize("kw", [dict(), dict(importance_mode="cascade"), dict(pq_threshold=0.05, head_keep=6), dict(fuse_qkv=True, native_gemv=True)])
def test_decode_graph_with_a_batch_of_two_sequences(kw):
    """B = 2 through the patched forward (the reference chats with one sequence; the boundary carries a batch): the captured
    loop equals the eager loop bit for bit, and so does the prune that follows."""
    from spatten_amd.graph import DecodeGraph
    dt = torch.bfloat16
    a, b, (cache_a, cache_b) = _models(dt, **kw)
    g = torch.Generator(device="cuda").manual_seed(17)
    x0 = torch.randn(2, 150, HID, device="cuda", generator=g).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=7)
        for t in range(7):
            x = torch.randn(2, 1, HID, device="cuda", generator=g).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            assert torch.equal(ya, yb), (turn, t)
        past_b = graph.past_key_values
        new_a = cache_a.apply_token_pruning(past_a, 12, [m.attn_scores for m in a.layers])
        new_b = cache_b.apply_token_pruning(past_b, 12, [m.attn_scores for m in b.layers])
        ext = getattr(cache_a, "ext", None)
        kept = [slice(None) if ext is None or st.head_ids is None else st.head_ids.long() for st in (ext.layers if ext else a.layers)]
        for (ka, va), (kb, vb), hk in zip(new_a, new_b, kept):
            assert ka.shape[0] == 2 and torch.equal(ka[:, hk], kb[:, hk]) and torch.equal(va[:, hk], vb[:, hk])
        xp = torch.randn(2, 5, HID, device="cuda", generator=g).to(dt)
        _, past_a = a(xp, new_a)
        _, past_b = b(xp, new_b)


@pytest.mark.parametrize("dt,d,kw", [(torch.float16, 64, dict(fuse_qkv=True, native_gemv=True)), (torch.bfloat16, 64, dict(pq_threshold=0.05)),
                                     (torch.bfloat16, 64, dict(importance_mode="cascade", head_keep=6)),
                                     (torch.float32, 128, dict(importance_mode="cascade")), (torch.float32, 64, dict(head_keep=6)),
                                     (torch.float16, 128, dict(pq_threshold=0.05, importance_mode="cascade")),
                                     (torch.float16, 128, dict(layer_keep=[50, 44, 44]))])
def test_decode_graph_other_head_dims_and_dtypes(monkeypatch, dt, d, kw):
    """head_dim 64 and the fp32 / f16 instantiations of the modes the graph covers, against the eager loop, bit for bit."""
    import sys

    from spatten_amd.graph import DecodeGraph
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "D", d)
    monkeypatch.setattr(mod, "HID", H * d)
    a, b, (cache_a, cache_b) = _models(dt, **kw)
    g = torch.Generator(device="cuda").manual_seed(23)
    x0 = torch.randn(1, 160, H * d, device="cuda", generator=g).to(dt)
    _, past_a = a(x0, None)
    _, past_b = b(x0, None)
    for turn in range(2):
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), past_b, horizon=6)
        for t in range(6):
            x = torch.randn(1, 1, H * d, device="cuda", generator=g).to(dt)
            ya, past_a = a(x, past_a)
            yb = graph.step(x)
            assert torch.equal(ya, yb), (turn, t)
        past_b = graph.past_key_values
        new_a = cache_a.apply_token_pruning(past_a, 11, [m.attn_scores for m in a.layers])
        new_b = cache_b.apply_token_pruning(past_b, 11, [m.attn_scores for m in b.layers])
        ext = getattr(cache_a, "ext", None)
        kept = [slice(None) if ext is None or st.head_ids is None else st.head_ids.long() for st in (ext.layers if ext else a.layers)]
        for (ka, va), (kb, vb), hk in zip(new_a, new_b, kept):
            assert torch.equal(ka[:, hk], kb[:, hk]) and torch.equal(va[:, hk], vb[:, hk])
        xp = torch.randn(1, 5, H * d, device="cuda", generator=g).to(dt)
        _, past_a = a(xp, new_a)
        _, past_b = b(xp, new_b)


def test_auto_graph_falls_back_when_the_step_cannot_be_captured():
    """A model whose single-token forward synchronises with the host (here: `.item()` on the logits) cannot be captured:
    auto_graph warns once, hands the call — and every later one — to the original forward, and the loop's results are the
    eager loop's."""
    import contextlib
    import io
    import warnings

    from spatten_amd import enable_spatten_llm
    dt = torch.bfloat16

    class _Syncing(_HFStyleLM):
        def forward(self, input_ids=None, past_key_values=None, use_cache=None, attention_mask=None, position_ids=None):
            out = _HFStyleLM.forward(self, input_ids, past_key_values, use_cache)
            self.last = out.logits[0, -1, 0].item()            # a host read inside the model call
            return out
    torch.manual_seed(5)
    a, b = _Syncing(dt), _Syncing(dt)
    b.load_state_dict(a.state_dict())
    for m, auto in ((a, False), (b, True)):
        with contextlib.redirect_stdout(io.StringIO()):
            enable_spatten_llm(m, 4, 40, 40, auto_graph=auto)
    ids = torch.randint(0, _TinyLM.VOCAB, (1, 60), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    with torch.no_grad(), warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        outs = []
        for m in (a, b):
            o = m(input_ids=ids, past_key_values=None, use_cache=True)
            past, tok, logits = o.past_key_values, o.logits[:, -1:].argmax(-1), []
            for _ in range(6):
                o = m(input_ids=tok, past_key_values=past, use_cache=True)
                past, tok = o.past_key_values, o.logits[:, -1:].argmax(-1)
                logits.append(o.logits.clone())
            outs.append((logits, past))
    assert any("auto_graph" in str(w.message) for w in caught) and b._spatten_auto_graph.get("disabled")
    assert all(torch.equal(x, y) for x, y in zip(outs[0][0], outs[1][0]))
    for (ka, va), (kb, vb) in zip(outs[0][1], outs[1][1]):
        assert torch.equal(ka, kb) and torch.equal(va, vb)
