"""Local V pruning as ONE launch (spatten_attn_decode_local_v, round 4) — PARITY UNPINNED (SpAttenController.scala:546-558,
591-612): against the r02 three-launch composition (scores-only decode + per-head top-k + gather P.V: same stash bit for bit,
same kept set) and against the oracle's restatement; the tie rule (lowest index first) across the splits of a head; the
device-length form."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.test_gpu_cascade import setup_decode
from tests.util import OUT_TOL, TORCH_DT, dev, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,d,B,H,Hkv,P,frac", [("bf16", 128, 1, 8, 8, 3000, 0.3), ("f16", 64, 2, 8, 4, 900, 0.5),
                                                  ("f32", 128, 1, 4, 4, 700, 0.1), ("bf16", 128, 1, 40, 40, 16383, 0.3),
                                                  ("bf16", 128, 2, 32, 32, 2080, 0.25), ("bf16", 64, 1, 2, 1, 40, 0.9)])
def test_one_launch_equals_the_three_launch_composition_and_the_oracle(dt, d, B, H, Hkv, P, frac):
    from spatten_amd import ops
    from spatten_amd.cascade import local_v_decode
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 51)
    keep = max(1, int(np.ceil(frac * N)))
    lse1 = torch.zeros(B, H, 2, dtype=torch.float32, device="cuda")
    lse3 = torch.zeros_like(lse1)
    o1, s1 = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep, lse=lse1)
    o3, s3 = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep, lse=lse3, three_launches=True)
    torch.cuda.synchronize()
    assert torch.equal(s1[:, :, :N], s3[:, :, :N])                       # the stash: same logits, same roundings
    np.testing.assert_allclose(lse1.cpu().numpy(), lse3.cpu().numpy(), atol=1e-5, rtol=1e-5)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == "f32" else OUT_TOL[dt]
    np.testing.assert_allclose(host(o1), host(o3), **tol)
    # the oracle on the SAME logits (the stash): probabilities of the full row, top-`keep`, ties lowest index first
    if N <= 4096:
        st = host(s1)[:, :, :N]
        want = orc.local_value_prune(orc.softmax_probs(st), orc.repeat_kv(vc, H // Hkv), keep)
        np.testing.assert_allclose(host(o1).reshape(B, H, d), want, **tol)


def test_ties_at_the_threshold_go_to_the_lowest_indices_across_splits():
    """All logits equal (a zero query): every key ties at the threshold, so the kept set must be exactly the first `keep`
    rows of the head — over several splits (split s keeps what the splits before it leave) — and out = mean-weighted sum."""
    from spatten_amd import ops
    dt, d, B, H, N = "bf16", 128, 1, 8, 6000
    g = torch.Generator(device="cuda").manual_seed(3)
    kr = torch.randn(B, H, N, d, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, N, d, device="cuda", generator=g).to(torch.bfloat16)
    q = torch.zeros(B, H, d, dtype=torch.bfloat16, device="cuda")
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    for keep in (1, 37, 2999, 3000, 3001, N - 1, N):
        stash = torch.empty(B, H, N, dtype=torch.bfloat16, device="cuda")
        out = ops.attn_decode_local_v(q, kr, v, N, cos, sin, N - 1, keep, stash)
        torch.cuda.synchronize()
        assert float(stash.float().abs().max()) == 0.0
        want = v[:, :, :keep].float().sum(2) / N
        np.testing.assert_allclose(out.float().cpu().numpy().reshape(B, H, d), want.cpu().numpy(), atol=2e-3, rtol=2e-2)


def test_mixed_ties_and_greater_keys_keep_exactly_keep_rows():
    """A two-valued logit row: `g` keys strictly above the threshold scattered over the splits, the rest tied — the kept set
    is the g greater rows plus the first (keep - g) tied rows, checked through a V plane that encodes the row index."""
    from spatten_amd import ops
    dt, d, B, H, N = "f32", 64, 1, 4, 5000
    rng = np.random.default_rng(9)
    base = np.zeros((B, H, N, d), np.float32)
    hot = [np.sort(rng.choice(N, size=300 + 50 * h, replace=False)) for h in range(H)]
    for h in range(H):
        base[0, h, hot[h], 0] = 4.0                     # q . k = 4 * q0 on the hot rows, 0 elsewhere (position 0: no rotation)
    q = np.zeros((B, H, d), np.float32)
    q[..., 0] = 1.0
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    # keys are given ALREADY "rotated" (kr_cache); the query is rotated at position 0 (identity rotation)
    v = np.zeros((B, H, N, d), np.float32)
    v[..., 0] = 1.0                                     # out[..., 0] = sum of kept probabilities
    v[..., 1] = np.arange(N)[None, None] / N            # out[..., 1] = sum p_j * j / N
    for keep in (100, 800, 2500):
        stash = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
        out = ops.attn_decode_local_v(dev(q, dt), dev(base, dt), dev(v, dt), N, cos, sin, 0, keep, stash)
        torch.cuda.synchronize()
        st = host(stash)
        probs = orc.softmax_probs(st)
        want = orc.local_value_prune(probs, v, keep)
        np.testing.assert_allclose(host(out).reshape(B, H, d), want, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_device_length_form_equals_the_static_launch_bitwise(dt):
    from spatten_amd import ops
    d, B, H, Hkv, P, frac = 128, 1, 8, 8, 2500, 0.35
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 52)
    cap = N + 200
    tdt = TORCH_DT[dt]
    kr2 = torch.zeros(B, Hkv, cap, d, dtype=tdt, device="cuda")
    v2 = torch.zeros_like(kr2)
    kr2[:, :, :N], v2[:, :, :N] = krd, vd
    c_p, s_p = orc.rope_table(cap, d, dt)
    cos_p, sin_p = dev(c_p[:, : d // 2], dt), dev(s_p[:, : d // 2], dt)
    st = ops.StepState(cos_p, sin_p)
    st.set(N - 1, N - 2)
    st.advance()
    keep = int(np.ceil(frac * N))
    sa = torch.zeros(B, H, cap, dtype=tdt, device="cuda")
    sb = torch.zeros_like(sa)
    la = torch.zeros(B, H, 2, dtype=torch.float32, device="cuda")
    lb = torch.zeros_like(la)
    oa = ops.attn_decode_local_v(qd, kr2, v2, N, cos_p, sin_p, N - 1, keep, sa, lse=la, layout=cap)
    ob = ops.attn_decode_local_v(qd, kr2, v2, cap, cos_p, sin_p, 0, 1, sb, lse=lb, keep_fraction=frac, step=st)
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(sa, sb) and torch.equal(la, lb)
    assert float(sb[:, :, N:].abs().max()) == 0.0


@pytest.mark.parametrize("dt,d,B,H,Hkv,P,frac", [("bf16", 128, 1, 8, 8, 2500, 0.35), ("f16", 64, 2, 8, 4, 700, 0.5),
                                                  ("f32", 128, 1, 4, 4, 300, 0.2), ("bf16", 128, 1, 40, 40, 16383, 0.3),
                                                  ("bf16", 128, 2, 4, 2, 100, 0.9), ("bf16", 128, 1, 2, 2, 1, 1.0),
                                                  ("f16", 128, 1, 32, 32, 4100, 0.25)])
def test_append_inside_the_launch_equals_append_then_launch(dt, d, B, H, Hkv, P, frac):
    """spatten_attn_decode_local_v_append (round 5): the step's append inside the one launch — cache rows and the stash bit for bit
    what spatten_kv_append followed by spatten_attn_decode_local_v leave, the same kept set; (max, sum) and the output agree to
    rounding (the appended key is folded first instead of last).  Host- and device-length forms (bitwise equal to each other)."""
    from spatten_amd import ops
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N0) = setup_decode(B, H, Hkv, d, P, dt, 58)
    N = N0 + 1                                   # the step appends row N0
    cap = N + 100
    tdt = TORCH_DT[dt]
    c_p, s_p = orc.rope_table(cap, d, dt)
    cos_p, sin_p = dev(c_p[:, : d // 2], dt), dev(s_p[:, : d // 2], dt)
    k_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    v_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    keep = max(1, int(np.ceil(frac * N)))

    def fresh():
        k0 = torch.zeros(B, Hkv, cap, d, dtype=tdt, device="cuda")
        kr0, v0 = torch.zeros_like(k0), torch.zeros_like(k0)
        kr0[:, :, :N0], v0[:, :, :N0] = krd, vd
        return k0, kr0, v0, torch.zeros(B, H, cap, dtype=tdt, device="cuda"), torch.zeros(B, H, 2, dtype=torch.float32, device="cuda")

    ka, kra, va, sa, la = fresh()
    ops.kv_append(k_new[:, :, None], v_new[:, :, None], ka, kra, va, N0, cos_p, sin_p)
    oa = ops.attn_decode_local_v(qd, kra, va, N, cos_p, sin_p, N - 1, keep, sa, lse=la, layout=cap)
    kb_, krb, vb, sb, lb = fresh()
    ob = ops.attn_decode_local_v(qd, krb, vb, N, cos_p, sin_p, N - 1, keep, sb, lse=lb, layout=cap, k_new=k_new, v_new=v_new, k_cache=kb_)
    kc_, krc, vc_, sc, lc = fresh()
    st = ops.StepState(cos_p, sin_p)
    st.set(N0, N0 - 1)
    st.advance()                                 # length N, query position N - 1
    oc = ops.attn_decode_local_v(qd, krc, vc_, cap, cos_p, sin_p, 0, 1, sc, lse=lc, keep_fraction=keep / N if False else frac, step=st,
                                 k_new=k_new, v_new=v_new, k_cache=kc_)
    torch.cuda.synchronize()
    assert torch.equal(ka, kb_) and torch.equal(kra, krb) and torch.equal(va, vb)
    assert torch.equal(sa, sb)
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), atol=1e-5, rtol=1e-5)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == "f32" else OUT_TOL[dt]
    np.testing.assert_allclose(host(oa), host(ob), **tol)
    # the device-length form: same rows, same stash; the kept count is ceil(frac * N) there
    assert torch.equal(ka, kc_) and torch.equal(kra, krc) and torch.equal(va, vc_) and torch.equal(sa, sc)
    if int(np.ceil(frac * N)) == keep:
        assert torch.equal(ob, oc) and torch.equal(lb, lc)


def test_more_units_than_cus_run_one_split_per_head_eager_and_device_length():
    """batch x heads above the CU count runs with ONE split per head (that path polls no sibling split) — the
    one-launch call must not be refused (-2) there, in its static form or in the device-length form a captured step uses."""
    from spatten_amd import ops
    from spatten_amd.cascade import local_v_decode
    dt, d, B, H, P = "bf16", 128, 9, 32, 600                   # 288 units > 256 CUs
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, H, d, P, dt, 77)
    keep = int(np.ceil(0.3 * N))
    st1 = torch.zeros(B, H, N, dtype=TORCH_DT[dt], device="cuda")
    o1 = ops.attn_decode_local_v(qd, krd, vd, N, cos, sin, N - 1, keep, st1)     # raises NotImplementedError if refused
    o3, s3 = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep, three_launches=True)
    torch.cuda.synchronize()
    assert torch.equal(st1[:, :, :N], s3[:, :, :N])
    np.testing.assert_allclose(host(o1), host(o3), **OUT_TOL[dt])
    # device-length form (what extensions.py's graph step launches): the same step with the length in the step state
    step = ops.StepState(cos, sin)
    step.set(N - 1, N - 2)
    step.advance()
    st2 = torch.zeros(B, H, krd.shape[2], dtype=TORCH_DT[dt], device="cuda")
    o2 = ops.attn_decode_local_v(qd, krd, vd, krd.shape[2], cos, sin, 0, 1, st2, keep_fraction=0.3, step=step)
    torch.cuda.synchronize()
    assert torch.equal(st2[:, :, :N], st1[:, :, :N])
    np.testing.assert_allclose(host(o2), host(o1), **OUT_TOL[dt])
