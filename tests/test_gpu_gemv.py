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
