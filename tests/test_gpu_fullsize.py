"""BASELINE.json full-size configurations (C3-C5 geometry) through the C ABI: oracle where it finishes in seconds,
size-independent properties elsewhere.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, TORCH_DT, check_stash, dev, host

pytestmark = pytest.mark.gpu


def test_c5_llama13b_long_context_decode_and_prune():
    """C5: H = 40, d = 128, N = 16384, bf16; start 4 / important 4092 / recent 4096."""
    from spatten_amd import SpAttenKVCache, ops
    dt, B, H, d, N = "bf16", 1, 40, 128, 16384
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(5)
    K = torch.randn(B, H, N + 128, d, device="cuda", generator=gen).to(tdt)
    V = torch.randn(B, H, N + 128, d, device="cuda", generator=gen).to(tdt)
    q = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    Kr = torch.zeros_like(K)
    ops.build_shadow(K, Kr, 0, N, cos, sin)
    stash = torch.empty(B, H, N, dtype=tdt, device="cuda")
    out = ops.attn_decode(q, None, Kr, V, N, cos, sin, N - 1, scores=stash)
    torch.cuda.synchronize()
    # oracle on the same (device-generated) inputs; past = rows [0, N-1), new row = row N-1
    Kh, Vh, qh = host(K[:, :, :N]), host(V[:, :, :N]), host(q)[:, :, None, :]
    o, st, _ = orc.attention_core(qh, Kh[:, :, N - 1:], Vh[:, :, N - 1:], Kh[:, :, :N - 1], Vh[:, :, :N - 1],
                                  np.full((B, 1), N - 1), None, dt)
    np.testing.assert_allclose(host(out)[:, None], o, **OUT_TOL[dt])
    check_stash(host(stash)[:, :, None], st, dt)
    # prune 16384 -> 8192, all properties + indices vs the oracle on the kernel's own stash
    cache = SpAttenKVCache(4, 4096, 4092)
    past = [(K[:, :, :N], V[:, :, :N])]
    new = cache.apply_token_pruning(past, 0, [stash[:, :, None, :]])
    torch.cuda.synchronize()
    idx = cache.keep_indices.cpu().numpy()[0]
    assert np.array_equal(idx, orc.topk_window(host(stash)[0], 4, N - 4096, 4092))
    Kn, Vn = new[0]
    assert Kn.shape == (B, H, 8192, d)
    ii = torch.from_numpy(idx).cuda().long()[:, :, None].expand(-1, -1, d)
    assert torch.equal(Kn[0, :, 4:4096], torch.gather(K[0, :, :N], 1, ii))
    assert torch.equal(Vn[0, :, 4:4096], torch.gather(V[0, :, :N], 1, ii))
    assert torch.equal(Kn[:, :, :4], K[:, :, :4]) and torch.equal(Vn[:, :, 4096:], V[:, :, N - 4096:N])
    # the shadow that came with the pruned slab == rotation of the new cache at its new slots (bit exact)
    from spatten_amd import kv_slab
    slab = Kn._spatten_slab
    tc, ts = kv_slab.rope_tables(slab.capacity, d, tdt, "cuda")        # the table the product path used
    want = ops.rope_single(Kn.contiguous(), tc, ts)
    assert torch.equal(slab.kr[:, :, :8192], want)


def test_c4_prefill_8192_properties_and_pq_planes():
    """C4: q = N = 8192 prefill (flash leg) + progressive-quant planes of its keys.  Properties: rows of a causal
    prefill equal the decode kernel on the same prefix; linearity in V; PQ with the LSB plane ~ 8-bit keys."""
    from spatten_amd import ops
    dt, B, H, d, N = "bf16", 1, 4, 128, 8192
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(4)
    K = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    V = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    Q = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    cos, sin = ops.rope_table(N, d, tdt, "cuda")
    Kr = ops.rope_single(K, cos, sin)
    out = ops.attn_prefill(Q, Kr, V, N, cos, sin, 0, causal=True)
    # row i of the causal prefill == decode of query i over keys [0, i]  (two different kernels, same answer)
    for i in (0, 1, 63, 64, 4097, N - 1):
        o1 = ops.attn_decode(Q[:, :, i].contiguous(), None, Kr, V, i + 1, cos, sin, i)
        np.testing.assert_allclose(host(out[:, i]), host(o1), atol=1e-2, rtol=2e-2)
    # linearity in V: attention(Q, K, 2V) == 2 attention(Q, K, V) (exact in bf16: power-of-two scale)
    out2 = ops.attn_prefill(Q, Kr, (V * 2).contiguous(), N, cos, sin, 0, causal=True)
    assert torch.equal(out2, out * 2)
    # progressive quantisation of the rotated keys
    planes = ops.PQPlanes(B, H, N, d, "cuda")
    ops.pq_pack(Kr, planes, 0, N)
    msb, lsb, scale = planes.unpack(N)
    deq = orc.pq_dequant(msb, lsb, scale)
    assert np.abs(deq - host(Kr)).max() <= 0.5 * scale.max() + 1e-6          # 8-bit symmetric: half a step
    q1 = Q[:, :, N - 1].contiguous()
    need = torch.zeros(B * H, dtype=torch.int32, device="cuda")
    o8 = ops.attn_decode_pq(q1, planes, V, N, cos, sin, N - 1, 2.0, need_lsb=need)     # always refetch
    o4 = ops.attn_decode_pq(q1, planes, V, N, cos, sin, N - 1, 0.0)                    # MSB plane only
    of = ops.attn_decode(q1, None, Kr, V, N, cos, sin, N - 1)
    assert need.all()
    e8 = float((o8.float() - of.float()).abs().max())
    e4 = float((o4.float() - of.float()).abs().max())
    assert e8 < 0.03 and e8 <= e4 + 1e-3, (e8, e4)


@pytest.mark.parametrize("seed", range(6))
def test_topk_random_shapes_vs_oracle(seed):
    from spatten_amd import ops
    rng = np.random.default_rng(seed)
    H = int(rng.integers(1, 9))
    L = int(rng.integers(2, 9000))
    lo = int(rng.integers(0, L - 1))
    hi = int(rng.integers(lo + 1, L + 1))
    k = int(rng.integers(1, hi - lo + 1))
    dt = ("f32", "bf16", "f16")[seed % 3]
    s = orc.round_dt((rng.standard_normal((H, L)) * 10 ** rng.uniform(-3, 3)).astype(np.float32), dt)
    assert np.array_equal(ops.topk_select(dev(s, dt), lo, hi, k).cpu().numpy(), orc.topk_window(s, lo, hi, k))
