"""Profiled progressive-quantisation planes (ABI 4): key MSB plane of 4 / 6 / 8 bits + 4-bit LSB plane, value plane of
8 / 6 bits, LSB-only refetch.  PARITY UNPINNED: vs the oracle's restatement (oracle/spatten_oracle.py: pq_quantize,
pq_quantize_values, pq_decode_attention_profile — MatrixFetcher.scala:48-51, TestSpAtten.scala:64,83-97,173-176,
SpAttenController.scala:35-39,402,716-723, RequantDecision.scala:44-72).  Bars: planes bit exact; refetch flags equal
wherever |max prob - threshold| > 1e-4; outputs within the model-dtype output tolerance; logits (stash) within one rounding
step of the model dtype."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.test_gpu_cascade import setup_decode
from tests.util import OUT_TOL, TORCH_DT, dev, host

pytestmark = pytest.mark.gpu

PROFILES = [(4, 8), (8, 8), (6, 6)]


def _oracle_planes(kr_host, v_host, kb, vb):
    msb, lsb, scale = orc.pq_quantize(kr_host, bits=kb + 4)
    qv, vscale = orc.pq_quantize_values(v_host, bits=vb)
    return msb.astype(np.int64), lsb.astype(np.int64), scale, qv.astype(np.int64), vscale


def _build(B, H, Hkv, d, P, dt, seed, kb, vb, extra=9):
    from spatten_amd import ops
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, seed)
    planes = ops.PQProfilePlanes(B, Hkv, H, N + extra, d, "cuda", key_bits=kb, value_bits=vb)
    ops.pq_pack_planes(krd, vd, planes, 0, N)
    want = _oracle_planes(host(krd), host(vd), kb, vb)
    c, s = orc.rope_table(N, d, dt)
    qr = orc.apply_rotary_pos_emb_single(q, c, s, np.full((B, 1), N - 1), dt)[:, :, 0]
    return planes, want, qr, qd, krd, vd, cos, sin, N


@pytest.mark.parametrize("kb,vb", PROFILES)
@pytest.mark.parametrize("dt,d,Hkv", [("bf16", 128, 8), ("f16", 64, 4), ("bf16", 64, 8)])
def test_profile_planes_bit_exact_and_decode_vs_oracle(kb, vb, dt, d, Hkv):
    from spatten_amd import ops
    B, H, P = 2, 8, 700
    planes, (msb, lsb, scale, qv, vscale), qr, qd, krd, vd, cos, sin, N = _build(B, H, Hkv, d, P, dt, 43, kb, vb)
    gm, gl, gs, gq, gvs = planes.unpack(N)
    assert np.array_equal(gm, msb) and np.array_equal(gl, lsb) and np.array_equal(gs, scale)
    assert np.array_equal(gq, qv) and np.array_equal(gvs, vscale)
    rep = lambda a: orc.repeat_kv(a, H // Hkv)
    p1max = orc.softmax_probs(np.einsum("bhd,bhld->bhl", qr.astype(np.float32), orc.pq_dequant(rep(msb), None, rep(scale)))
                              / np.float32(np.sqrt(d))).max(-1)
    for thr in (0.0, 2.0, float(np.median(p1max))):
        want, need, logits = orc.pq_decode_attention_profile(qr, rep(msb), rep(lsb), rep(scale), rep(qv), rep(vscale), thr)
        need_dev = torch.full((B * H,), -1, dtype=torch.int32, device="cuda")
        scores = torch.zeros(B, H, N + 9, dtype=TORCH_DT[dt], device="cuda")
        lse = torch.zeros(B, H, 2, dtype=torch.float32, device="cuda")
        out = ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, thr, need_lsb=need_dev, scores=scores, lse=lse)
        torch.cuda.synchronize()
        got_need = need_dev.cpu().numpy().reshape(B, H).astype(bool)
        decided = np.abs(p1max - thr) > 1e-4
        assert np.array_equal(got_need[decided], need[decided]), (thr, got_need, need)
        ok = got_need == need
        np.testing.assert_allclose(host(out).reshape(B, H, d)[ok], orc.round_dt(want, dt)[ok], **OUT_TOL[dt])
        # the stash holds the logits the output was computed from, rounded to the model dtype
        st = host(scores)[:, :, :N]
        tol = 2.0 ** (-7 if dt == "bf16" else -10)
        assert np.all(np.abs(st - logits)[ok] <= tol * np.maximum(np.abs(logits[ok]), 1.0) + 1e-6)
        # (max, sum) of the row the output was computed from
        m = logits.max(-1)
        np.testing.assert_allclose(lse[..., 0].cpu().numpy()[ok], m[ok], atol=1e-4, rtol=1e-5)
    # sanity: 8-bit-ish keys and values stay close to the un-quantised attention
    full = ops.attn_decode(qd, None, krd, vd, N, cos, sin, N - 1)
    outq = ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, 2.0)
    assert float((full.float() - outq.float()).abs().max()) < (0.12 if vb == 6 else 0.05)


@pytest.mark.parametrize("kb,vb", PROFILES)
def test_profile_decode_splits_head_subsets_and_device_length(kb, vb):
    """Long rows (several splits and pipelined tiles), a kept-head list, head importance, and the device-length form: equal to
    the static launch bit for bit (same split layout)."""
    from spatten_amd import ops
    dt, d, B, H, Hkv, P = "bf16", 128, 1, 8, 8, 5000
    planes, (msb, lsb, scale, qv, vscale), qr, qd, krd, vd, cos, sin, N = _build(B, H, Hkv, d, P, dt, 44, kb, vb, extra=64)
    thr = 0.5
    want, need, logits = orc.pq_decode_attention_profile(qr, msb, lsb, scale, qv, vscale, thr)
    out = ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, thr)
    np.testing.assert_allclose(host(out).reshape(B, H, d), orc.round_dt(want, dt), **OUT_TOL[dt])
    # forced split counts agree with each other to rounding
    for S in (1, 3, 16):
        o2 = ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, thr, n_splits=S)
        np.testing.assert_allclose(host(o2), host(out), **OUT_TOL[dt])
    # kept-head list: only those heads are written
    keep = torch.tensor([1, 2, 5, 7], dtype=torch.int32, device="cuda")
    o3 = torch.full((B, H * d), float("nan"), dtype=TORCH_DT[dt], device="cuda")
    ha = torch.zeros(B * H, dtype=torch.float32, device="cuda")
    ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, thr, out=o3, head_ids=keep, head_abs=ha)
    o3 = o3.view(B, H, d)
    for h in range(H):
        if h in (1, 2, 5, 7):
            assert torch.equal(o3[:, h], out.view(B, H, d)[:, h])
            assert abs(float(ha[h]) - float(o3[0, h].float().abs().sum())) < 1e-2
        else:
            assert torch.isnan(o3[:, h].float()).all() and float(ha[h]) == 0.0
    # device-length form: bound = capacity, length from the step state; same layout as the static launch laid out for it
    cap = planes.capacity
    c_p, s_p = orc.rope_table(cap, d, dt)
    cos_p, sin_p = dev(c_p[:, : d // 2], dt), dev(s_p[:, : d // 2], dt)
    st = ops.StepState(cos_p, sin_p)
    st.set(N - 1, N - 2)
    st.advance()
    sc_a = torch.zeros(B, H, cap, dtype=TORCH_DT[dt], device="cuda")
    sc_b = torch.zeros_like(sc_a)
    na = torch.zeros(B * H, dtype=torch.int32, device="cuda")
    nb = torch.zeros_like(na)
    o_static = ops.attn_decode_pqv(qd, planes, N, cos_p, sin_p, N - 1, thr, scores=sc_a, need_lsb=na, layout=cap)
    o_dyn = ops.attn_decode_pqv(qd, planes, cap, cos_p, sin_p, 0, thr, scores=sc_b, need_lsb=nb, step=st)
    torch.cuda.synchronize()
    assert torch.equal(o_static, o_dyn) and torch.equal(sc_a[:, :, :N], sc_b[:, :, :N]) and torch.equal(na, nb)
    assert float(sc_b[:, :, N:].abs().max()) == 0.0


@pytest.mark.parametrize("kb,vb", PROFILES)
def test_profile_pack_of_the_step_row_in_device_length_form(kb, vb):
    """spatten_pq_pack_planes(step_state): the one row (state length) - 1, equal to the host-length pack of that row."""
    from spatten_amd import ops
    dt, d, B, H, Hkv, P = "f16", 128, 2, 4, 2, 300
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 45)
    a = ops.PQProfilePlanes(B, Hkv, H, N, d, "cuda", key_bits=kb, value_bits=vb)
    b = ops.PQProfilePlanes(B, Hkv, H, N, d, "cuda", key_bits=kb, value_bits=vb)
    ops.pq_pack_planes(krd, vd, a, 0, N)
    ops.pq_pack_planes(krd, vd, b, 0, N - 1)
    st = ops.StepState(cos, sin)
    st.set(N - 1, N - 2)
    st.advance()
    ops.pq_pack_planes(krd, vd, b, 0, N, step=st)
    torch.cuda.synchronize()
    for x, y in ((a.msb, b.msb), (a.lsb, b.lsb), (a.scale, b.scale), (a.vq, b.vq), (a.vscale, b.vscale)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("kb,vb", PROFILES)
@pytest.mark.parametrize("dt,d,Hkv", [("bf16", 128, 8), ("f16", 64, 4), ("f32", 128, 2)])
def test_fused_append_and_plane_rows_equal_the_two_launches(kb, vb, dt, d, Hkv):
    """spatten_kv_append_planes (round 5): the step's append and its plane rows in ONE launch — cache rows and every plane bit for
    bit what spatten_kv_append followed by spatten_pq_pack_planes leave, in the host-length and in the device-length form."""
    from spatten_amd import ops
    B, H, P = 2, 8, 300
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 46)
    cap = N + 8
    tdt = TORCH_DT[dt]
    k_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    v_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    c_p, s_p = orc.rope_table(cap, d, dt)
    cos_p, sin_p = dev(c_p[:, : d // 2], dt), dev(s_p[:, : d // 2], dt)

    def fresh():
        k0 = torch.zeros(B, Hkv, cap, d, device="cuda", dtype=tdt)
        kr0 = torch.zeros_like(k0)
        v0 = torch.zeros_like(k0)
        kr0[:, :, :N] = krd
        v0[:, :, :N] = vd
        pl = ops.PQProfilePlanes(B, Hkv, H, cap, d, "cuda", key_bits=kb, value_bits=vb)
        ops.pq_pack_planes(kr0, v0, pl, 0, N)
        return k0, kr0, v0, pl

    ka, kra, va, pa = fresh()
    ops.kv_append(k_new[:, :, None], v_new[:, :, None], ka, kra, va, N, cos_p, sin_p)
    ops.pq_pack_planes(kra, va, pa, N, N + 1)
    kb_, krb, vb_, pb = fresh()
    ops.kv_append_planes(k_new, v_new, kb_, krb, vb_, pb, N, cos_p, sin_p)
    kc_, krc, vc_, pc = fresh()
    st = ops.StepState(cos_p, sin_p)
    st.set(N, N - 1)
    st.advance()                               # length N + 1: the step's row is N
    ops.kv_append_planes(k_new, v_new, kc_, krc, vc_, pc, 0, cos_p, sin_p, step=st)
    torch.cuda.synchronize()
    for other_k, other_kr, other_v, other_p in ((kb_, krb, vb_, pb), (kc_, krc, vc_, pc)):
        assert torch.equal(ka, other_k) and torch.equal(kra, other_kr) and torch.equal(va, other_v)
        for x, y in ((pa.msb, other_p.msb), (pa.lsb, other_p.lsb), (pa.scale, other_p.scale), (pa.vq, other_p.vq),
                     (pa.vscale, other_p.vscale)):
            assert torch.equal(x, y)
    with pytest.raises(ValueError):
        ops.kv_append_planes(k_new, v_new, kb_, krb, vb_, pb, cap, cos_p, sin_p)


@pytest.mark.parametrize("kb,vb", PROFILES)
@pytest.mark.parametrize("dt,d,H,Hkv,P", [("bf16", 128, 8, 8, 5000), ("f16", 64, 8, 4, 300), ("f32", 128, 4, 2, 130), ("bf16", 128, 2, 2, 1)])
def test_append_inside_the_msb_pass_equals_append_then_decode(kb, vb, dt, d, H, Hkv, P):
    """spatten_pq_decode_args_t.k_new / v_new (round 5): the step's append and its plane rows inside the MSB pass — cache rows, every
    plane, the stash and the refetch flags bit for bit what spatten_kv_append_planes followed by the decode leave; (max, sum) and the
    output agree to rounding (the appended key is folded first instead of last).  Host- and device-length forms; a head list appends
    only for the heads it launches."""
    from spatten_amd import ops
    B = 2
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N0) = setup_decode(B, H, Hkv, d, P, dt, 47)
    N, cap = N0 + 1, N0 + 1 + 40
    tdt = TORCH_DT[dt]
    k_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    v_new = torch.randn(B, Hkv, d, device="cuda", dtype=torch.float32).to(tdt)
    c_p, s_p = orc.rope_table(cap, d, dt)
    cos_p, sin_p = dev(c_p[:, : d // 2], dt), dev(s_p[:, : d // 2], dt)
    thr = 0.02

    def fresh():
        k0 = torch.zeros(B, Hkv, cap, d, device="cuda", dtype=tdt)
        kr0, v0 = torch.zeros_like(k0), torch.zeros_like(k0)
        kr0[:, :, :N0], v0[:, :, :N0] = krd, vd
        pl = ops.PQProfilePlanes(B, Hkv, H, cap, d, "cuda", key_bits=kb, value_bits=vb)
        ops.pq_pack_planes(kr0, v0, pl, 0, N0)
        return (k0, kr0, v0, pl, torch.zeros(B, H, cap, dtype=tdt, device="cuda"), torch.zeros(B, H, 2, dtype=torch.float32, device="cuda"),
                torch.full((B * H,), -1, dtype=torch.int32, device="cuda"))

    def same_state(x, y, heads_kv=None):
        sel = slice(None) if heads_kv is None else heads_kv
        for a_, b_ in ((x[0], y[0]), (x[1], y[1]), (x[2], y[2]), (x[3].msb, y[3].msb), (x[3].lsb, y[3].lsb), (x[3].scale, y[3].scale),
                       (x[3].vq, y[3].vq), (x[3].vscale, y[3].vscale)):
            assert torch.equal(a_[:, sel], b_[:, sel])

    A = fresh()
    ops.kv_append_planes(k_new, v_new, A[0], A[1], A[2], A[3], N0, cos_p, sin_p)
    oa = ops.attn_decode_pqv(qd, A[3], N, cos_p, sin_p, N - 1, thr, scores=A[4], lse=A[5], need_lsb=A[6], layout=cap)
    Bf = fresh()
    ob = ops.attn_decode_pqv(qd, Bf[3], N, cos_p, sin_p, N - 1, thr, scores=Bf[4], lse=Bf[5], need_lsb=Bf[6], layout=cap,
                             append=(k_new, v_new, Bf[0], Bf[1], Bf[2]))
    C = fresh()
    st = ops.StepState(cos_p, sin_p)
    st.set(N0, N0 - 1)
    st.advance()
    oc = ops.attn_decode_pqv(qd, C[3], cap, cos_p, sin_p, 0, thr, scores=C[4], lse=C[5], need_lsb=C[6], step=st,
                             append=(k_new, v_new, C[0], C[1], C[2]))
    torch.cuda.synchronize()
    same_state(A, Bf)
    same_state(A, C)
    assert torch.equal(A[4], Bf[4]) and torch.equal(A[6], Bf[6])
    assert torch.equal(Bf[4], C[4]) and torch.equal(Bf[6], C[6]) and torch.equal(ob, oc) and torch.equal(Bf[5], C[5])
    np.testing.assert_allclose(A[5].cpu().numpy(), Bf[5].cpu().numpy(), atol=1e-5, rtol=1e-5)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == "f32" else OUT_TOL[dt]
    np.testing.assert_allclose(host(oa), host(ob), **tol)
    # a head list: only the launched heads' kv heads get the row
    if H == Hkv and H >= 4:
        keep = torch.tensor([1, 2], dtype=torch.int32, device="cuda")
        Dd = fresh()
        o_d = torch.zeros(B, H * d, dtype=tdt, device="cuda")
        ops.attn_decode_pqv(qd, Dd[3], N, cos_p, sin_p, N - 1, thr, out=o_d, scores=Dd[4], need_lsb=Dd[6], layout=cap, head_ids=keep,
                            append=(k_new, v_new, Dd[0], Dd[1], Dd[2]))
        torch.cuda.synchronize()
        same_state(A, Dd, heads_kv=[1, 2])
        assert float(Dd[1][:, [0, 3], N0].abs().max()) == 0.0
        np.testing.assert_allclose(host(o_d.view(B, H, d)[:, [1, 2]]), host(ob.view(B, H, d)[:, [1, 2]]), **tol)   # (other split count)


def test_profile_c4_c5_scale_decode_vs_oracle():
    """BASELINE.json configs[3] / configs[4] geometry through the profiled planes: Llama-2-7B heads at 8192 rows (4, 8) and
    Llama-2-13B heads (H = 40) at 16384 rows (8, 8) — outputs and refetch flags vs the oracle."""
    from spatten_amd import ops
    for (H, P, kb, vb, seed) in ((32, 8191, 4, 8, 46), (40, 16383, 8, 8, 47)):
        dt, d, B = "bf16", 128, 1
        planes, (msb, lsb, scale, qv, vscale), qr, qd, krd, vd, cos, sin, N = _build(B, H, H, d, P, dt, seed, kb, vb, extra=0)
        p1max = orc.softmax_probs(np.einsum("bhd,bhld->bhl", qr.astype(np.float32), orc.pq_dequant(msb, None, scale))
                                  / np.float32(np.sqrt(d))).max(-1)
        thr = float(np.median(p1max))
        want, need, _ = orc.pq_decode_attention_profile(qr, msb, lsb, scale, qv, vscale, thr)
        need_dev = torch.zeros(B * H, dtype=torch.int32, device="cuda")
        out = ops.attn_decode_pqv(qd, planes, N, cos, sin, N - 1, thr, need_lsb=need_dev)
        torch.cuda.synchronize()
        got = need_dev.cpu().numpy().reshape(B, H).astype(bool)
        decided = np.abs(p1max - thr) > 1e-4
        assert np.array_equal(got[decided], need[decided])
        ok = got == need
        np.testing.assert_allclose(host(out).reshape(B, H, d)[ok], orc.round_dt(want, dt)[ok], **OUT_TOL[dt])
        assert 0 < need.sum() < need.size


def test_c5_geometry_at_a_trace_like_refetch_rate_flags_exactly_the_oracles_heads():
    """the reference's traces refetch 323 of 4,608 head-requests (~7 %:
    workloads/summary-gpt2-small-wikitext2-per8.csv, columns auto_requant_thres / if_requant).  C5 geometry (H = 40, d = 128,
    8192 rows, (8, 8) planes, threshold 0.05) with queries PEAKED on the newest key for 37 of the 40 heads (bench.py
    --pq-confidence trace builds its inputs the same way): the MSB pass must flag exactly the 3 unpeaked heads — the set the
    oracle flags — refetch only those, and leave the outputs within tolerance."""
    from spatten_amd import ops
    from tests.util import attn_inputs
    dt, d, B, H, P, kb, vb, thr = "bf16", 128, 1, 40, 8191, 8, 8, 0.05
    q, k, v, past = attn_inputs(B, H, H, d, P + 1, 1, dt, 61)
    kc, vc = past
    N = P + 1
    flagged = (5, 17, 33)                                               # 3 of 40 = 7.5 %
    for h in range(H):
        if h not in flagged:      # the newest key = 0.9 x the query: relative position 0, logit ~ 0.9 |q|^2 / sqrt(d) ~ 10
            kc[:, h, N - 1] = orc.round_dt(np.float32(0.9) * q[:, h, 0], dt)
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    kd, vd = dev(kc, dt), dev(vc, dt)
    krd = ops.rope_single(kd, cos, sin)
    planes = ops.PQProfilePlanes(B, H, H, N, d, "cuda", key_bits=kb, value_bits=vb)
    ops.pq_pack_planes(krd, vd, planes, 0, N)
    msb, lsb, scale, qv, vscale = _oracle_planes(host(krd), host(vd), kb, vb)
    qr = orc.apply_rotary_pos_emb_single(q, c, s, np.full((B, 1), N - 1), dt)[:, :, 0]
    want, need, _ = orc.pq_decode_attention_profile(qr, msb, lsb, scale, qv, vscale, thr)
    assert sorted(np.nonzero(need[0])[0].tolist()) == list(flagged)      # the oracle: exactly the unpeaked heads
    need_dev = torch.full((B * H,), -1, dtype=torch.int32, device="cuda")
    out = ops.attn_decode_pqv(dev(q[:, :, 0], dt), planes, N, cos, sin, N - 1, thr, need_lsb=need_dev)
    torch.cuda.synchronize()
    got = need_dev.cpu().numpy().reshape(B, H).astype(bool)
    assert np.array_equal(got, need)
    np.testing.assert_allclose(host(out).reshape(B, H, d), orc.round_dt(want, dt), **OUT_TOL[dt])
