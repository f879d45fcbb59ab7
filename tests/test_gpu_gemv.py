"""The single-token projection kernel (spatten_gemv; modify_llama.py:72-74, :163 at q_len = 1) against a plain PyTorch
fp32 reference of nn.Linear, and through the patched forward.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from tests.util import host

pytestmark = pytest.mark.gpu

# fp32 accumulation + ONE rounding to the dtype, like torch's linear; only the summation order differs
TOL = {torch.float32: dict(atol=1e-4, rtol=1e-5), torch.bfloat16: dict(atol=2e-2, rtol=8e-3), torch.float16: dict(atol=3e-3, rtol=1e-3)}


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("M,N,K,bias", [(1, 4096, 4096, False), (1, 12288, 4096, True), (3, 5120, 5120, True),
                                        (1, 37, 520, True), (2, 1000, 8, False), (1, 16, 11008, False)])
def test_gemv_matches_an_fp32_linear(dt, M, N, K, bias):
    from spatten_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    b = torch.randn(N, device="cuda", generator=g).to(dt) if bias else None
    y = ops.gemv(x, W, b)
    torch.cuda.synchronize()
    want = torch.nn.functional.linear(x.float(), W.float(), None if b is None else b.float())
    assert y.shape == (M, N) and y.dtype == dt
    np.testing.assert_allclose(host(y), host(want), **TOL[dt])
    # a [B, 1, K] activation and a row-strided weight view (the stacked q/k/v weight's slices)
    big = torch.zeros(N + 8, K, dtype=dt, device="cuda")
    big[4:4 + N] = W
    y3 = ops.gemv(x[:, None, :], big[4:4 + N], b)
    assert y3.shape == (M, 1, N) and torch.equal(y3[:, 0], y)


def test_gemv_rejects_what_it_does_not_cover():
    from spatten_amd import ops
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemv(z(1, 12), z(4, 12))                 # K not a multiple of 8
    with pytest.raises(ValueError):
        ops.gemv(z(1, 16), z(4, 24))
    with pytest.raises(RuntimeError):
        ops.gemv(torch.zeros(1, 16), torch.zeros(4, 16))     # CPU tensors: no fallback


@pytest.mark.parametrize("B,H,Hkv,N,bias,fused", [(1, 32, 32, 2081, False, True), (1, 32, 32, 300, True, True), (1, 16, 16, 1500, False, True),
                                                   (2, 32, 32, 700, True, False), (1, 32, 8, 900, False, False), (1, 4, 4, 5000, True, False)])
def test_decode_step_with_its_output_projection_equals_the_two_launches(B, H, Hkv, N, bias, fused):
    """proj_* of the decode argument block: the step's o_proj (modify_llama.py:163) issued by the attention call must equal
    spatten_gemv on the attention output bit for bit, in the static and the device-length form, and leave the attention
    output and stash unchanged."""
    from spatten_amd import ops
    tdt, d = torch.bfloat16, 128
    g = torch.Generator(device="cuda").manual_seed(N + H)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(tdt)
    cap = (N + 64 + 127) // 128 * 128
    cos, sin = ops.rope_table(cap, d, tdt, "cuda")
    K = torch.zeros(B, Hkv, cap, d, dtype=tdt, device="cuda")
    V = torch.zeros_like(K)
    K[:, :, :N - 1], V[:, :, :N - 1] = rnd(B, Hkv, N - 1, d), rnd(B, Hkv, N - 1, d)
    Kr = torch.zeros_like(K)
    ops.build_shadow(K, Kr, 0, N - 1, cos, sin)
    q, kn, vn = rnd(B, H, d), rnd(B, Hkv, d), rnd(B, Hkv, d)
    W = (torch.randn(H * d + 24, H * d, device="cuda", generator=g) * (H * d) ** -0.5).to(tdt)
    b_ = rnd(W.shape[0]) if bias else None
    ref = {}
    for mode in ("two launches", "static", "device length"):
        k2, kr2, v2 = K.clone(), Kr.clone(), V.clone()
        stash = torch.zeros(B, H, cap, dtype=tdt, device="cuda")
        y = torch.full((B, W.shape[0]), float("nan"), dtype=tdt, device="cuda")
        if mode == "two launches":
            out = ops.attn_decode(q, k2, kr2, v2, N, cos, sin, N - 1, k_new=kn, v_new=vn, scores=stash, layout=cap)
            y = ops.gemv(out, W, b_)
        elif mode == "static":
            out = ops.attn_decode(q, k2, kr2, v2, N, cos, sin, N - 1, k_new=kn, v_new=vn, scores=stash, layout=cap, proj=(W, b_, y))
        else:
            st = ops.StepState(cos, sin)
            st.set(N - 1, N - 2)
            st.advance()
            out = ops.attn_decode(q, k2, kr2, v2, cap, cos, sin, 0, k_new=kn, v_new=vn, scores=stash, step=st, proj=(W, b_, y))
        torch.cuda.synchronize()
        assert not torch.isnan(y.float()).any(), mode
        if not ref:
            ref = dict(out=out.clone(), y=y.clone(), stash=stash.clone())
        else:
            assert torch.equal(out, ref["out"]) and torch.equal(stash[:, :, :N], ref["stash"][:, :, :N]), mode
            assert torch.equal(y, ref["y"]), (mode, (y.float() - ref["y"].float()).abs().max().item())
    want = torch.nn.functional.linear(ref["out"].float(), W.float(), None if b_ is None else b_.float())
    np.testing.assert_allclose(host(ref["y"]), host(want), **TOL[tdt])
    # several such calls back to back on one workspace, with plain launches in between
    if fused:
        ws = ops.DecodeWorkspace(B, H, d, "cuda")
        ys = []
        for rep in range(5):
            k2, kr2, v2 = K.clone(), Kr.clone(), V.clone()
            y = torch.empty(B, W.shape[0], dtype=tdt, device="cuda")
            ops.attn_decode(q, k2, kr2, v2, N, cos, sin, N - 1, k_new=kn, v_new=vn, layout=cap, proj=(W, b_, y), workspace=ws)
            if rep % 2:       # an unfused launch in between must not disturb the counters
                ops.attn_decode(q, k2, kr2, v2, N, cos, sin, N - 1, layout=cap, workspace=ws)
            ys.append(y)
        torch.cuda.synchronize()
        ws.check()
        for y in ys:
            assert torch.equal(y, ref["y"])
