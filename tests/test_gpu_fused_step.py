"""The layer-step's q / k / v projections INSIDE the attention launch (decode_qkv_kernel, round 4; modify_llama.py:72-74 +
:86-147): bit-identical to spatten_gemv + the plain fused decode step — output, stash, appended cache rows (un-rotated key,
rotated shadow, value) — in the static and the device-length form, at several split counts and hidden sizes; and through the
plugin (enable_spatten_llm(fused_step=True)): the eager loop runs the fused launch and equals the unfused plugin bit for bit,
DecodeGraph replays on the same model too (the device-length form of the fused launch is covered at the op level above)."""
import contextlib
import io
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

from oracle import spatten_oracle as orc
from tests.util import TORCH_DT, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _team_256():
    """The fused launch contains the 256-thread attention team: bit-identity with the separate step is stated against that
    team (enable_spatten_llm(fused_step=True) selects it; the op-level tests do it here)."""
    from spatten_amd import ops
    prev = ops.set_decode_team(256)
    yield
    ops.set_decode_team(prev)


def _planes(H, cap, d, P, tdt, g):
    k = torch.zeros(1, H, cap, d, dtype=tdt, device="cuda")
    v = torch.zeros_like(k)
    kr = torch.zeros_like(k)
    k[:, :, :P] = (torch.randn(1, H, P, d, device="cuda", generator=g) * d ** -0.5).to(tdt)
    v[:, :, :P] = torch.randn(1, H, P, d, device="cuda", generator=g).to(tdt)
    return k, kr, v


@pytest.mark.parametrize("dt,H,hidden,P,S", [("bf16", 32, 4096, 2047, 0), ("f16", 32, 4096, 1500, 0), ("bf16", 8, 1024, 900, 8),
                                             ("bf16", 8, 1000, 600, 4), ("f16", 4, 6144, 300, 2), ("bf16", 16, 2048, 200, 1),
                                             ("bf16", 32, 4096, 2400, 0)])
def test_fused_projection_step_equals_gemv_plus_the_plain_step_bitwise(dt, H, hidden, P, S):
    from spatten_amd import ops
    d, tdt = 128, TORCH_DT[dt]
    g = torch.Generator(device="cuda").manual_seed(7)
    cap = P + 1 + 60
    N = P + 1
    c, s = orc.rope_table(cap, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    ka, kra, va = _planes(H, cap, d, P, tdt, g)
    ops.build_shadow(ka, kra, 0, P, cos, sin)
    kb, krb, vb = ka.clone(), kra.clone(), va.clone()
    x = torch.randn(1, 1, hidden, device="cuda", generator=g).to(tdt)
    w = (torch.randn(3 * H * d, hidden, device="cuda", generator=g) * hidden ** -0.5).to(tdt)
    bias = (torch.randn(3 * H * d, device="cuda", generator=g) * 0.1).to(tdt) if H == 4 else None
    wsa, wsb = ops.DecodeWorkspace(1, H, d, "cuda"), ops.DecodeWorkspace(1, H, d, "cuda")
    # (a) the separate launches: spatten_gemv, then the plain step
    qkv = ops.gemv(x, w, bias).view(3, H, d)
    sa = torch.zeros(1, H, cap, dtype=tdt, device="cuda")
    oa = ops.attn_decode(qkv[0][None], ka, kra, va, N, cos, sin, P, k_new=qkv[1][None], v_new=qkv[2][None], scores=sa,
                         n_splits=S, workspace=wsa, layout=cap)
    # (b) one launch
    sb = torch.zeros_like(sa)
    ob = ops.attn_decode_qkv(x, w, bias, H, kb, krb, vb, N, cos, sin, P, scores=sb, n_splits=S, workspace=wsb, layout=cap)
    torch.cuda.synchronize()
    assert torch.equal(oa, ob), float((oa.float() - ob.float()).abs().max())
    assert torch.equal(sa, sb)
    assert torch.equal(ka, kb) and torch.equal(kra, krb) and torch.equal(va, vb)
    wsb.check()
    # a second step on the same planes (the exchange granules are re-tagged by the launch generation), device-length form
    st = ops.StepState(cos, sin)
    st.set(N, N - 1)
    st.advance()
    x2 = torch.randn(1, 1, hidden, device="cuda", generator=g).to(tdt)
    qkv2 = ops.gemv(x2, w, bias).view(3, H, d)
    oa2 = ops.attn_decode(qkv2[0][None], ka, kra, va, N + 1, cos, sin, N, k_new=qkv2[1][None], v_new=qkv2[2][None], scores=sa,
                          n_splits=S, workspace=wsa, layout=cap)
    ob2 = ops.attn_decode_qkv(x2, w, bias, H, kb, krb, vb, cap, cos, sin, 0, scores=sb, n_splits=S, workspace=wsb, step=st)
    torch.cuda.synchronize()
    assert torch.equal(oa2, ob2) and torch.equal(sa[:, :, :N + 1], sb[:, :, :N + 1])
    assert torch.equal(ka, kb) and torch.equal(kra, krb) and torch.equal(va, vb)


@pytest.mark.parametrize("dt,with_bias", [("bf16", False), ("f16", True)])
def test_output_projection_inside_the_fused_launch_equals_spatten_gemv_bitwise(dt, with_bias):
    """Llama-2-7B geometry (32 heads x 128, 8 splits = 256 workgroups x 16 o_proj rows): the whole attention module of a decode
    step as ONE launch; y equals spatten_gemv(o_proj) on the separate step's output, over several steps on one workspace (the
    output exchange has its own generation word), static and device-length form."""
    from spatten_amd import ops
    H, d, hidden, P, tdt = 32, 128, 4096, 2047, TORCH_DT[dt]
    g = torch.Generator(device="cuda").manual_seed(11)
    cap = P + 1 + 60
    c, s = orc.rope_table(cap, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    ka, kra, va = _planes(H, cap, d, P, tdt, g)
    ops.build_shadow(ka, kra, 0, P, cos, sin)
    kb, krb, vb = ka.clone(), kra.clone(), va.clone()
    w = (torch.randn(3 * H * d, hidden, device="cuda", generator=g) * hidden ** -0.5).to(tdt)
    wo = (torch.randn(hidden, H * d, device="cuda", generator=g) * hidden ** -0.5).to(tdt)
    bo = (torch.randn(hidden, device="cuda", generator=g) * 0.1).to(tdt) if with_bias else None
    wsa, wsb = ops.DecodeWorkspace(1, H, d, "cuda"), ops.DecodeWorkspace(1, H, d, "cuda")
    sa = torch.zeros(1, H, cap, dtype=tdt, device="cuda")
    sb = torch.zeros_like(sa)
    st = ops.StepState(cos, sin)
    st.set(P, P - 1)
    for t in range(4):
        N = P + 1 + t
        x = torch.randn(1, 1, hidden, device="cuda", generator=g).to(tdt)
        qkv = ops.gemv(x, w, None).view(3, H, d)
        oa = ops.attn_decode(qkv[0][None], ka, kra, va, N, cos, sin, N - 1, k_new=qkv[1][None], v_new=qkv[2][None], scores=sa,
                             workspace=wsa, layout=cap)
        ya = ops.gemv(oa.view(1, 1, H * d), wo, bo)
        st.advance()
        if t % 2 == 0:
            ob, yb = ops.attn_decode_qkv(x, w, None, H, kb, krb, vb, N, cos, sin, N - 1, scores=sb, workspace=wsb, layout=cap,
                                         proj=(wo, bo))
        else:
            ob, yb = ops.attn_decode_qkv(x, w, None, H, kb, krb, vb, cap, cos, sin, 0, scores=sb, workspace=wsb, step=st,
                                         proj=(wo, bo))
        torch.cuda.synchronize()
        assert torch.equal(oa, ob)
        assert torch.equal(ya.view(-1), yb.view(-1)), (t, float((ya.float().view(-1) - yb.float().view(-1)).abs().max()))
        assert torch.equal(sa[:, :, :N], sb[:, :, :N])
        wsb.check()
    # the same call with the in-launch projection switched off (SPATTEN_FUSED_OPROJ=0 is read once per process: compare
    # against a geometry the in-launch form does not cover instead — 8 heads: one spatten_gemv launch behind the step)
    H2 = 8
    k2, kr2, v2 = _planes(H2, cap, d, 600, tdt, g)
    ops.build_shadow(k2, kr2, 0, 600, cos, sin)
    x2 = torch.randn(1, 1, 1024, device="cuda", generator=g).to(tdt)
    w2 = (torch.randn(3 * H2 * d, 1024, device="cuda", generator=g) * 1024 ** -0.5).to(tdt)
    wo2 = (torch.randn(1024, H2 * d, device="cuda", generator=g) * 1024 ** -0.5).to(tdt)
    s2 = torch.zeros(1, H2, cap, dtype=tdt, device="cuda")
    o2, y2 = ops.attn_decode_qkv(x2, w2, None, H2, k2, kr2, v2, 601, cos, sin, 600, scores=s2, n_splits=8, layout=cap, proj=(wo2, None))
    assert torch.equal(y2.view(-1), ops.gemv(o2.view(1, 1, H2 * d), wo2).view(-1))


def test_fused_step_refuses_the_shapes_it_does_not_cover():
    from spatten_amd import _lib, ops
    lib = _lib.load()
    assert lib.spatten_decode_qkv_supported(2, 1, 32, 32, 128, 2432) == 1          # Llama-2-7B heads on a 2432-row slab
    assert lib.spatten_decode_qkv_supported(2, 1, 32, 32, 128, 4096) == 0          # longer than a single-shot tile per split
    assert lib.spatten_decode_qkv_supported(2, 1, 40, 40, 128, 2048) == 0          # 6 splits: not a divisor of the head
    assert lib.spatten_decode_qkv_supported(0, 1, 32, 32, 128, 2048) == 0          # fp32
    assert lib.spatten_decode_qkv_supported(2, 1, 32, 8, 128, 2048) == 0           # grouped-query
    assert lib.spatten_decode_qkv_supported(2, 2, 32, 32, 128, 2048) == 0          # batch 2


H, D, LAYERS = 32, 128, 2
HID = H * D


class LlamaAttention(nn.Module):          # duck-typed by class name, like HF's module
    def __init__(self, dt):
        super().__init__()
        self.config = SimpleNamespace(pretraining_tp=1)
        self.num_heads = self.num_key_value_heads = H
        self.num_key_value_groups, self.head_dim, self.hidden_size = 1, D, HID
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin = nn.Linear(HID, HID, bias=False, dtype=dt, device="cuda")
            nn.init.normal_(lin.weight, std=HID ** -0.5)
            setattr(self, n, lin)


class Stack(nn.Module):
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


def test_plugin_with_fused_step_equals_the_unfused_plugin_bitwise_eager_and_under_decode_graph():
    from spatten_amd import enable_spatten_llm, kv_slab
    from spatten_amd.graph import DecodeGraph
    dt = torch.bfloat16
    torch.manual_seed(0)
    a = Stack(dt)
    b = Stack(dt)
    b.load_state_dict(a.state_dict())
    with contextlib.redirect_stdout(io.StringIO()):
        ca = enable_spatten_llm(a, 4, 60, 64, fuse_qkv=True, native_gemv=True, assume_causal=True)
        cb = enable_spatten_llm(b, 4, 60, 64, fuse_qkv=True, native_gemv=True, assume_causal=True, fused_step=True)
    g = torch.Generator(device="cuda").manual_seed(2)
    P, T = 300, 10
    x0 = torch.randn(1, P, HID, device="cuda", generator=g).to(dt)
    _, pa = a(x0, None)
    _, pb = b(x0, None)
    pa = kv_slab.reserve(pa, P + 3 * T + 8)          # equal slab capacities: equal split layouts (HISTORY.md, r04 §3.8)
    pb = kv_slab.reserve(pb, P + 3 * T + 8)
    launches = []
    from spatten_amd import ops
    orig = ops.SlabDecodeCall._run_qkv

    def counted(self, *args, **kw):
        launches.append(1)
        return orig(self, *args, **kw)
    ops.SlabDecodeCall._run_qkv = counted
    try:
        for t in range(T):                            # eager loop
            x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
            ya, pa = a(x, pa)
            yb, pb = b(x, pb)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), t
        assert len(launches) == T * LAYERS            # the fused launch DID run
        for ma, mb in zip(a.layers, b.layers):
            assert torch.equal(ma.attn_scores, mb.attn_scores)
        n_eager = len(launches)
        graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), pb, horizon=T)
        for t in range(T):                            # graph replays (a traced step projects first: the separate launches are
                                                      # the faster form under a graph, and bit-identical) vs the eager unfused loop
            x = torch.randn(1, 1, HID, device="cuda", generator=g).to(dt)
            ya, pa = a(x, pa)
            yb = graph.step(x)
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), t
        assert graph.n_replays == T - 1 and len(launches) == n_eager
        pb = graph.past_key_values
    finally:
        ops.SlabDecodeCall._run_qkv = orig
    for (ka, va), (kb, vb), ma, mb in zip(pa, pb, a.layers, b.layers):
        assert torch.equal(ka, kb) and torch.equal(va, vb) and torch.equal(ma.attn_scores, mb.attn_scores)
        assert torch.equal(kv_slab.slab_of(ka).kr[:, :, :ka.shape[2]], kv_slab.slab_of(kb).kr[:, :, :kb.shape[2]])
    # the prune event that follows is the same
    na = ca.apply_token_pruning(pa, 40, [m.attn_scores for m in a.layers])
    nb = cb.apply_token_pruning(pb, 40, [m.attn_scores for m in b.layers])
    for (ka, va), (kb, vb) in zip(na, nb):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


def test_fused_step_warns_about_the_process_wide_team_and_can_restore_it():
    """fused_step=True selects the 256-thread decode team for the WHOLE process — it says so, and the cache object
    can put the previous team back."""
    import warnings

    from spatten_amd import enable_spatten_llm, ops
    ops.set_decode_team(512)
    m = Stack(torch.bfloat16)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cache = enable_spatten_llm(m, 4, 60, 64, fuse_qkv=True, native_gemv=True, assume_causal=True, fused_step=True)
    assert any("256-thread decode team" in str(x.message) for x in w)
    assert ops.set_decode_team(256) == 256          # selected
    cache.restore_process_options()
    assert ops.set_decode_team(512) == 512          # restored
