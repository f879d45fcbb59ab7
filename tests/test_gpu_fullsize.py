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


def _pq_setup(B, H, d, N, dt, seed, peaky_rows=()):
    """Un-rotated Q (queries over rows [P, N)), rotated shadow Kr, V on the device + the planes; `peaky_rows` get a query
    aligned with one key so that their attention is confident (max prob above any small threshold)."""
    from spatten_amd import ops
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(seed)
    K = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    V = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    cos, sin = ops.rope_table(N, d, tdt, "cuda")
    Kr = ops.rope_single(K, cos, sin)
    planes = ops.PQPlanes(B, H, N, d, "cuda")
    ops.pq_pack(Kr, planes, 0, N)
    return K, V, Kr, planes, cos, sin, gen


@pytest.mark.parametrize("dt,d,P,ql", [("bf16", 128, 0, 700), ("f16", 64, 333, 300), ("bf16", 128, 1000, 260)])
def test_prefill_pq_vs_oracle(dt, d, P, ql):
    """PQ-keyed prefill (configs[3] semantics) vs the oracle's restatement, every row: MSB-pass flags per query row,
    refetched rows recomputed from the 8-bit keys, P.V with the un-quantised V."""
    from spatten_amd import ops
    B, H, N = 2, 4, P + ql
    tdt = TORCH_DT[dt]
    K, V, Kr, planes, cos, sin, gen = _pq_setup(B, H, d, N, dt, 11)
    Q = (torch.randn(B, H, ql, d, device="cuda", generator=gen) * 1.5).to(tdt)
    msb, lsb, scale = planes.unpack(N)
    c, s = host(cos), host(sin)
    cs, sn = np.concatenate([c, c], -1), np.concatenate([s, s], -1)
    qr = orc.apply_rotary_pos_emb_single(host(Q), cs, sn, np.arange(P, N)[None], dt)
    _, _, pmax = orc.pq_prefill_attention(qr, msb, lsb, scale, host(V), 0.0, P)
    for thr in (0.0, 2.0, float(np.median(pmax))):
        want, need, _ = orc.pq_prefill_attention(qr, msb, lsb, scale, host(V), thr, P)
        out, need_g = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, P, thr, causal=True)
        torch.cuda.synchronize()
        ng = need_g.cpu().numpy().astype(bool)
        clear = np.abs(pmax - thr) > 1e-4 * max(thr, 1e-3)          # rows whose decision is not a rounding coin-flip
        assert np.array_equal(ng[clear], need[clear]), thr
        assert 0 < clear.mean()
        ok_rows = clear.all(axis=1)                                  # [B, q]: compare rows decided alike in every head
        got = host(out).reshape(B, ql, H, d)
        np.testing.assert_allclose(got[ok_rows], orc.round_dt(want, dt).reshape(B, ql, H, d)[ok_rows], **OUT_TOL[dt])
    # threshold 0 = MSB keys only, threshold 2 = every row refetched = plain 8-bit keys: the refetch brings the result
    # closer to the un-quantised attention
    o4, _ = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, P, 0.0)
    o8, n8 = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, P, 2.0)
    of = ops.attn_prefill(Q, Kr, V, N, cos, sin, P, causal=True)
    assert bool(n8.all())
    e4, e8 = float((o4.float() - of.float()).abs().max()), float((o8.float() - of.float()).abs().max())
    assert e8 < 0.05 and e8 <= e4 + 1e-3, (e4, e8)


def test_c4_prefill_8192_pq_keyed_full_size():
    """configs[3] AS WRITTEN: Llama-2-7B geometry (H = 32, d = 128), q = N = 8192 causal prefill over progressively
    quantised keys, bf16.  Sampled query rows against the oracle (every head), refetch flags for all of them; full-size
    properties: a row of the PQ prefill equals the PQ decode step on the same prefix, pass 2 only touches flagged rows."""
    from spatten_amd import ops
    dt, B, H, d, N = "bf16", 1, 32, 128, 8192
    K, V, Kr, planes, cos, sin, gen = _pq_setup(B, H, d, N, dt, 44)
    Q = torch.randn(B, H, N, d, device="cuda", generator=gen).to(TORCH_DT[dt])
    # make a band of rows confident: query row i of the band points at key i (its own position) with a large norm
    band = torch.arange(4000, 4400, device="cuda")
    Q[:, :, band] = (K[:, :, band].float() * 6.0).to(Q.dtype)
    thr = 0.05                                                      # the traces' auto_requant_thres
    out, need = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, 0, thr, causal=True)
    torch.cuda.synchronize()
    need_h = need.cpu().numpy().astype(bool)
    assert 0.02 < need_h.mean() < 0.999 and not need_h[:, :, 4100:4300].all()      # a mix: both passes matter
    rows = [0, 1, 127, 128, 2047, 4000, 4100, 4200, 4399, 4400, 6000, N - 1]
    msb, lsb, scale = planes.unpack(N)
    c, s = host(cos), host(sin)
    cs, sn = np.concatenate([c, c], -1), np.concatenate([s, s], -1)
    qr = orc.apply_rotary_pos_emb_single(host(Q[:, :, rows]), cs, sn, np.asarray(rows)[None], dt)
    want = np.zeros((B, len(rows), H * d), np.float32)
    for n, i in enumerate(rows):                                    # oracle row by row (the prefix of row i only)
        w, nd, pm = orc.pq_prefill_attention(qr[:, :, n:n + 1], msb[:, :, :i + 1], lsb[:, :, :i + 1], scale[:, :, :i + 1],
                                             host(V[:, :, :i + 1]), thr, i)
        want[:, n] = w[:, 0]
        clear = np.abs(pm[:, :, 0] - thr) > 1e-4
        assert np.array_equal(need_h[:, :, i][clear], nd[:, :, 0][clear]), i
        if clear.all():
            np.testing.assert_allclose(host(out[:, i]), orc.round_dt(want[:, n], dt), err_msg=f"row {i}", **OUT_TOL[dt])
    # the same row through the decode kernel's PQ path (per-HEAD decision there: compare heads that agree)
    for i in (2047, 4200, N - 1):
        nd = torch.empty(B * H, dtype=torch.int32, device="cuda")
        o1 = ops.attn_decode_pq(Q[:, :, i].contiguous(), planes, V, i + 1, cos, sin, i, thr, need_lsb=nd)
        assert torch.equal(nd.view(B, H) != 0, need[:, :, i] != 0)
        np.testing.assert_allclose(host(out[:, i]), host(o1), atol=1e-2, rtol=2e-2)
    # rows that were not flagged carry exactly the MSB-pass result (pass 2 left them alone)
    o_msb, _ = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, 0, 0.0, causal=True)
    keep = (need == 0).permute(0, 2, 1)                              # [B, q, H]
    assert torch.equal(out.view(B, N, H, d)[keep], o_msb.view(B, N, H, d)[keep])


@pytest.mark.parametrize("B,H,P,ql", [(2, 5, 300, 2000), (1, 8, 0, 2176)])
def test_prefill_pq_split_refetch_pass_ragged_every_row(B, H, P, ql):
    """The compacted, key-split refetch pass on ragged shapes (round 5), EVERY row against the oracle: head count not a multiple of
    8, a query block on a longer cache, key tiles that do not divide by the ranges; head 0 is flagged almost everywhere (it keeps
    the unsplit list pass: more than a quarter of its rows), the other heads have every 13th row flagged (split pass + merge)."""
    from spatten_amd import ops
    dt, d, N = "bf16", 128, P + ql
    K, V, Kr, planes, cos, sin, gen = _pq_setup(B, H, d, N, dt, 91)
    Q = (K[:, :, P:].float() * 6.0).to(TORCH_DT[dt])
    Q[:, :, 5::13] = torch.randn(B, H, len(range(5, ql, 13)), d, device="cuda", generator=gen).to(TORCH_DT[dt])
    Q[:, 0] = torch.randn(B, ql, d, device="cuda", generator=gen).to(TORCH_DT[dt])
    thr = 0.05
    msb, lsb, scale = planes.unpack(N)
    c, s = host(cos), host(sin)
    cs, sn = np.concatenate([c, c], -1), np.concatenate([s, s], -1)
    qr = orc.apply_rotary_pos_emb_single(host(Q), cs, sn, np.arange(P, N)[None], dt)
    want, need, pmax = orc.pq_prefill_attention(qr, msb, lsb, scale, host(V), thr, P)
    out, need_g = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, P, thr, causal=True)
    torch.cuda.synchronize()
    ng = need_g.cpu().numpy().astype(bool)
    frac = ng.mean(axis=-1)                                          # [B, H]
    assert (frac[:, 0] > 0.25).all() and (frac[:, 1:] < 0.25).all() and (frac[:, 1:] > 0.02).all(), frac
    clear = np.abs(pmax - thr) > 1e-4
    assert np.array_equal(ng[clear], need[clear])
    ok_rows = clear.all(axis=1)
    assert ok_rows.mean() > 0.9
    got = host(out).reshape(B, ql, H, d)
    np.testing.assert_allclose(got[ok_rows], orc.round_dt(want, dt).reshape(B, ql, H, d)[ok_rows], **OUT_TOL[dt])


def test_c4_prefill_8192_pq_few_flagged_rows_take_the_compacted_key_split_pass():
    """configs[3] at a realistic refetch rate (round 5): when at most a quarter of the query rows is flagged, pass 2 walks the
    COMPACTED list of flagged rows (256 list entries per workgroup) with its key tiles split over workgroups and a merge launch.
    Most rows here are confident (they point at their own key), every 19th is random: ~5 % flagged, scattered over all blocks."""
    from spatten_amd import ops
    dt, B, H, d, N = "bf16", 1, 32, 128, 8192
    K, V, Kr, planes, cos, sin, gen = _pq_setup(B, H, d, N, dt, 45)
    Q = (K.float() * 6.0).to(TORCH_DT[dt])
    loose = torch.arange(7, N, 19, device="cuda")
    Q[:, :, loose] = torch.randn(B, H, loose.numel(), d, device="cuda", generator=gen).to(TORCH_DT[dt])
    thr = 0.05
    out, need = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, 0, thr, causal=True)
    torch.cuda.synchronize()
    need_h = need.cpu().numpy().astype(bool)
    frac = need_h.mean(axis=-1)
    assert 0.01 < frac.min() and frac.max() < 0.25, (frac.min(), frac.max())      # every head takes the split pass
    rows = [7, 26, 8 + 19 * 100, 7 + 19 * 215, 7 + 19 * 400, 7 + 19 * 430, 4100, 8000]
    msb, lsb, scale = planes.unpack(N)
    c, s = host(cos), host(sin)
    cs, sn = np.concatenate([c, c], -1), np.concatenate([s, s], -1)
    qr = orc.apply_rotary_pos_emb_single(host(Q[:, :, rows]), cs, sn, np.asarray(rows)[None], dt)
    checked = 0
    for n, i in enumerate(rows):
        w, nd, pm = orc.pq_prefill_attention(qr[:, :, n:n + 1], msb[:, :, :i + 1], lsb[:, :, :i + 1], scale[:, :, :i + 1],
                                             host(V[:, :, :i + 1]), thr, i)
        clear = np.abs(pm[:, :, 0] - thr) > 1e-4
        assert np.array_equal(need_h[:, :, i][clear], nd[:, :, 0][clear]), i
        if clear.all():
            np.testing.assert_allclose(host(out[:, i]), orc.round_dt(w[:, 0], dt), err_msg=f"row {i}", **OUT_TOL[dt])
            checked += int(nd.any())
    assert checked >= 3                                                            # flagged rows were among the compared ones
    # un-flagged rows carry exactly the MSB-pass result; a flagged row equals the PQ decode step over its prefix
    o_msb, _ = ops.attn_prefill_pq(Q, planes, V, N, cos, sin, 0, 0.0, causal=True)
    keep = (need == 0).permute(0, 2, 1)
    assert torch.equal(out.view(B, N, H, d)[keep], o_msb.view(B, N, H, d)[keep])
    for i in (7 + 19 * 215, 7 + 19 * 430):
        nd = torch.empty(B * H, dtype=torch.int32, device="cuda")
        o1 = ops.attn_decode_pq(Q[:, :, i].contiguous(), planes, V, i + 1, cos, sin, i, thr, need_lsb=nd)
        assert torch.equal(nd.view(B, H) != 0, need[:, :, i] != 0)
        np.testing.assert_allclose(host(out[:, i]), host(o1), atol=1e-2, rtol=2e-2)


def test_c3_head_pruned_decode_full_size():
    """configs[2] at full size: Llama-2-7B geometry, 4096 -> 2048 token prune, head importance -> keep 24 of 32 heads ->
    decode launched on the kept heads only; everything against the oracle."""
    from spatten_amd import SpAttenKVCache, ops
    from spatten_amd.cascade import HeadPruner
    dt, B, H, d, N = "bf16", 1, 32, 128, 4096
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(33)
    K = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    V = (torch.randn(B, H, N, d, device="cuda", generator=gen) *
         torch.linspace(0.5, 2.0, H, device="cuda")[None, :, None, None]).to(tdt)       # heads of different weight
    q = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    c, s = orc.rope_table(N + 64, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    Kr = ops.rope_single(K, cos, sin)
    stash = torch.empty(B, H, N, dtype=tdt, device="cuda")
    head_abs = torch.zeros(B * H, dtype=torch.float32, device="cuda")
    out = ops.attn_decode(q, None, Kr, V, N, cos, sin, N - 1, scores=stash, head_abs=head_abs)
    torch.cuda.synchronize()
    # head importance fused into the decode launch == the oracle's sum |attn_out_h| == the separate kernel
    want_abs = orc.head_scores(host(out)[:, None, :], H)
    np.testing.assert_allclose(host(head_abs), want_abs, rtol=1e-5)
    hp = HeadPruner(H, "cuda")
    hp.observe(out[:, None, :])
    np.testing.assert_allclose(host(hp.scores), want_abs, rtol=1e-5)
    keep = hp.select(24)
    assert np.array_equal(keep.cpu().numpy(), orc.head_prune_select(want_abs, 24))
    # token prune 4096 -> 1984 (+ room for the 64 coming tokens; indices bit exact vs the oracle on the kernel's stash)
    cache = SpAttenKVCache(4, 1024, 1020)
    new = cache.apply_token_pruning([(K, V)], 64, [stash[:, :, None, :]])
    idx = cache.keep_indices.cpu().numpy()[0]
    assert np.array_equal(idx, orc.topk_window(host(stash)[0], 4, N - 1024 + 64, 1020))
    Kn, Vn = new[0]
    n = Kn.shape[2]
    assert n == 4 + 1020 + 960
    slab = Kn._spatten_slab
    # decode over the pruned cache, kept heads only, appending a token: vs the oracle on the pruned cache
    qn = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    kn = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    vn = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    Kh, Vh = host(Kn), host(Vn)
    o, st, _ = orc.attention_core(host(qn)[:, :, None], host(kn)[:, :, None], host(vn)[:, :, None], Kh, Vh,
                                  np.full((B, 1), n), None, dt)
    o2 = torch.zeros(B, H * d, dtype=tdt, device="cuda")
    st2 = torch.zeros(B, H, n + 1, dtype=tdt, device="cuda")
    tc, ts = slab.tables(n + 1)
    # (the product path's torch-built table; the oracle's numpy table differs in a few 16-bit entries -> tolerance only)
    ops.attn_decode(qn, slab.k, slab.kr, slab.v, n + 1, tc, ts, n, k_new=kn, v_new=vn, out=o2, scores=st2, head_ids=keep)
    torch.cuda.synchronize()
    kept = keep.cpu().numpy()
    dead = np.setdiff1d(np.arange(H), kept)
    np.testing.assert_allclose(host(o2).reshape(B, H, d)[:, kept], o.reshape(B, H, d)[:, kept], **OUT_TOL[dt])
    assert not host(o2).reshape(B, H, d)[:, dead].any() and not host(st2)[:, dead].any()     # pruned heads: untouched
    assert float(np.mean(host(st2)[:, kept] != st[:, kept, 0])) < 0.03


def test_c5_combined_token_head_prune_and_pq_decode_full_size():
    """configs[4] at full size: Llama-2-13B geometry (H = 40), 16384 -> 8192 token prune, head prune to 30 of 40,
    progressive-quant decode over the kept rows of the kept heads — vs the oracle."""
    from spatten_amd import SpAttenKVCache, ops
    dt, B, H, d, N = "bf16", 1, 40, 128, 16384
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(55)
    K = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    V = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    q = torch.randn(B, H, d, device="cuda", generator=gen).to(tdt)
    cos, sin = ops.rope_table(N, d, tdt, "cuda")
    Kr = ops.rope_single(K, cos, sin)
    stash = torch.empty(B, H, N, dtype=tdt, device="cuda")
    head_abs = torch.zeros(B * H, dtype=torch.float32, device="cuda")
    out = ops.attn_decode(q, None, Kr, V, N, cos, sin, N - 1, scores=stash, head_abs=head_abs)
    keep = ops.topk_select(head_abs[None, :], 0, H, 30)[0].contiguous()
    assert np.array_equal(keep.cpu().numpy(), orc.head_prune_select(orc.head_scores(host(out)[:, None, :], H), 30))
    cache = SpAttenKVCache(4, 4096, 4092)
    new = cache.apply_token_pruning([(K, V)], 0, [stash[:, :, None, :]])
    assert np.array_equal(cache.keep_indices.cpu().numpy()[0], orc.topk_window(host(stash)[0], 4, N - 4096, 4092))
    Kn, Vn = new[0]
    slab = Kn._spatten_slab
    n = 8192
    slab.ensure_pq(n)
    msb, lsb, scale = slab.pq.unpack(n)
    gm, gl, gs = orc.pq_quantize(host(slab.kr[:, :, :n]))
    assert np.array_equal(msb, gm) and np.array_equal(lsb, gl) and np.array_equal(scale, gs)
    tc, ts = slab.tables(n)
    c2, s2 = host(tc), host(ts)
    qn = (torch.randn(B, H, d, device="cuda", generator=gen) * 2).to(tdt)
    qr = orc.apply_rotary_pos_emb_single(host(qn)[:, :, None], np.concatenate([c2, c2], -1), np.concatenate([s2, s2], -1),
                                         np.full((B, 1), n - 1), dt)[:, :, 0]
    logits, _ = orc.pq_logits(qr, msb, lsb, scale, 0.0)
    thr = float(np.median(orc.softmax_probs(logits).max(-1)))
    want, need = orc.pq_decode_attention(qr, msb, lsb, scale, host(Vn), thr)
    o2 = torch.zeros(B, H * d, dtype=tdt, device="cuda")
    nd = torch.zeros(B * H, dtype=torch.int32, device="cuda")
    ops.attn_decode_pq(qn, slab.pq, slab.v, n, tc, ts, n - 1, thr, out=o2, need_lsb=nd, head_ids=keep)
    torch.cuda.synchronize()
    kept = keep.cpu().numpy()
    dead = np.setdiff1d(np.arange(H), kept)
    assert np.array_equal(nd.cpu().numpy().reshape(B, H).astype(bool)[:, kept], need[:, kept])
    np.testing.assert_allclose(host(o2).reshape(B, H, d)[:, kept], orc.round_dt(want, dt)[:, kept], **OUT_TOL[dt])
    assert not host(o2).reshape(B, H, d)[:, dead].any()


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
