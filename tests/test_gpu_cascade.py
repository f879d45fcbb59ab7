"""Cascade importance, local V pruning, head pruning (PARITY UNPINNED: vs the oracle's restatement only)."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, TORCH_DT, attn_inputs, dev, host

pytestmark = pytest.mark.gpu


def setup_decode(B, H, Hkv, d, P, dt, seed):
    from spatten_amd import ops
    q, k, v, past = attn_inputs(B, H, Hkv, d, P + 1, 1, dt, seed)   # treat all P+1 rows as cached
    kc, vc = past
    N = P + 1
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    kd, vd = dev(kc, dt), dev(vc, dt)
    krd = ops.rope_single(kd, cos, sin)
    pos = np.full((B, 1), N - 1)
    qr = orc.apply_rotary_pos_emb_single(q, c, s, pos, dt)
    kr = orc.repeat_kv(orc.apply_rotary_pos_emb_single(kc, c, s, np.arange(N)[None], dt), H // Hkv)
    stash = orc.round_dt(orc.round_dt(np.matmul(qr, np.swapaxes(kr, 2, 3)), dt) / np.float32(np.sqrt(d)), dt)
    return q, kc, vc, stash, (dev(q[:, :, 0], dt), krd, vd, cos, sin, N)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_cascade_importance_accumulate_and_compact(dt):
    from spatten_amd import ops
    from spatten_amd.cascade import CascadeImportance
    B, H, d, P = 2, 4, 64, 300
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, H, d, P, dt, 11)
    ci = CascadeImportance(1, H, 512, "cuda")
    st = torch.empty(B, H, N, dtype=TORCH_DT[dt], device="cuda")
    lse = torch.empty(B, H, 2, dtype=torch.float32, device="cuda")
    ops.attn_decode(qd, None, krd, vd, N, cos, sin, N - 1, scores=st, lse=lse)
    acc_want = np.zeros((H, N), np.float32)
    for rep in range(3):       # three steps accumulate
        ci.accumulate(0, st[:, :, None, :], lse[:, :, None, :])
        acc_want = orc.cascade_importance_accumulate(acc_want, host(st)[:, :, None, :])
    np.testing.assert_allclose(host(ci.acc[0])[:, :N], acc_want, rtol=2e-3, atol=1e-5)
    # lse recomputed from the stash (prefill-style stash with q > 1 rows and a causal mask)
    ql = 5
    st2 = dev(orc.synth_normal(3, 0, (B, H, ql, N), dt), dt)
    acc2 = torch.zeros(H, N, dtype=torch.float32, device="cuda")
    ops.importance_accumulate(acc2, st2, causal=True)
    mask = orc.causal_mask(B, ql, N, "f32")
    mask = np.where(mask < 0, -np.inf, 0).astype(np.float32)
    want2 = orc.cascade_importance_accumulate(np.zeros((H, N), np.float32), host(st2), mask)
    np.testing.assert_allclose(host(acc2), want2, rtol=2e-3, atol=1e-5)
    # selection on the accumulated importance + compaction of the accumulator (bit exact moves)
    idx = ci.select(0, N, 4, N - 50, 100)
    assert np.array_equal(idx.cpu().numpy(), orc.topk_window(host(ci.acc[0])[:, :N], 4, N - 50, 100))
    before = host(ci.acc[0])[:, :N]
    ci.compact(0, idx, 4, N - 50, N)
    ii = idx.cpu().numpy()
    want = np.concatenate([before[:, :4], np.take_along_axis(before, ii, 1), before[:, N - 50:]], axis=1)
    assert np.array_equal(host(ci.acc[0])[:, :want.shape[1]], want)


@pytest.mark.parametrize("dt,B", [("bf16", 1), ("f32", 2)])
def test_cascade_accumulation_fused_into_the_decode_step(dt, B):
    """Cumulative importance without a second kernel: each decode launch folds the PREVIOUS step's softmax probabilities
    (that step's stash + (max, sum)) into the accumulator while its own keys stream; the last step is folded by
    spatten_importance_accumulate at the prune.  Result = the oracle's running sum over all steps."""
    from spatten_amd import ops
    H, d, P, steps = 4, 128, 700, 5
    qs, ks, vs, past = attn_inputs(B, H, H, d, P, steps, dt, seed=17)     # `steps` new tokens after P cached ones
    cap = P + steps + 3
    kc = torch.zeros(B, H, cap, d, dtype=TORCH_DT[dt], device="cuda")
    krc, vc = torch.zeros_like(kc), torch.zeros_like(kc)
    kc[:, :, :P], vc[:, :, :P] = dev(past[0], dt), dev(past[1], dt)
    c, s = orc.rope_table(cap, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    ops.build_shadow(kc, krc, 0, P, cos, sin)
    acc = torch.zeros(H, cap, dtype=torch.float32, device="cuda")
    want = np.zeros((H, cap), np.float32)
    stash = [torch.zeros(B, H, cap, dtype=TORCH_DT[dt], device="cuda") for _ in range(2)]
    lse = [torch.zeros(B, H, 2, dtype=torch.float32, device="cuda") for _ in range(2)]
    pk, pv = past
    for t in range(steps):
        n = P + t + 1
        cur, prv = t & 1, (t & 1) ^ 1
        ops.attn_decode(dev(qs[:, :, t], dt), kc, krc, vc, n, cos, sin, n - 1, k_new=dev(ks[:, :, t], dt),
                        v_new=dev(vs[:, :, t], dt), scores=stash[cur], lse=lse[cur],
                        cascade=(acc, stash[prv], lse[prv], n - 1) if t > 0 else None)
        _, st, (pk, pv) = orc.attention_core(qs[:, :, t:t + 1], ks[:, :, t:t + 1], vs[:, :, t:t + 1], pk, pv,
                                             np.full((B, 1), n - 1), None, dt)
        torch.cuda.synchronize()
        # the oracle accumulates from the KERNEL's stash (16-bit stashes may differ by an ulp in < 2 % of entries)
        want[:, :n] = orc.cascade_importance_accumulate(want[:, :n], host(stash[cur][:, :, :n])[:, :, None, :])
        if t > 0:     # everything up to the previous step is in the accumulator after this launch
            prev_want = want[:, :n].copy()
            prev_want -= orc.softmax_probs(host(stash[cur][:, :, :n])[:, :, None, :]).sum(axis=(0, 2))
            np.testing.assert_allclose(host(acc)[:, :n], prev_want, rtol=2e-3, atol=2e-5)
    n = P + steps
    ops.importance_accumulate(acc, stash[(steps - 1) & 1][:, :, None, :n], lse[(steps - 1) & 1][:, :, None, :])   # the flush
    np.testing.assert_allclose(host(acc)[:, :n], want[:, :n], rtol=2e-3, atol=2e-5)
    assert float(host(acc)[:, n:].max()) == 0.0


@pytest.mark.parametrize("dt,d,Hkv,P,ql", [("bf16", 128, 4, 100, 600), ("f16", 64, 2, 0, 333), ("bf16", 128, 4, 900, 130)])
def test_cascade_importance_of_a_prefill_without_the_stash(dt, d, Hkv, P, ql):
    """Multi-token forwards in cascade mode: the flash kernel writes its row statistics, a second matrix-core pass
    recomputes the logits and sums the probabilities per key — no [B,H,q,N] stash.  vs the oracle's restatement (and vs
    the stash-based accumulation of the same launch)."""
    from spatten_amd import ops
    B, H, N = 2, 4, P + ql
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=23)
    kc = k if past is None else np.concatenate([past[0], k], 2)
    vc = v if past is None else np.concatenate([past[1], v], 2)
    c, s = orc.rope_table(N, d, dt)
    cos, sin = dev(c[:, : d // 2], dt), dev(s[:, : d // 2], dt)
    kd, vd, qd = dev(kc, dt), dev(vc, dt), dev(q, dt)
    krd = ops.rope_single(kd, cos, sin)
    lse = torch.empty(B, H, ql, 2, dtype=torch.float32, device="cuda")
    stash = torch.empty(B, H, ql, N, dtype=TORCH_DT[dt], device="cuda")
    out = ops.attn_prefill(qd, krd, vd, N, cos, sin, P, causal=True, lse=lse, scores=stash)
    out2 = ops.attn_prefill(qd, krd, vd, N, cos, sin, P, causal=True, lse=torch.empty_like(lse))      # 128-key-tile kernel
    assert torch.allclose(out.float(), out2.float(), atol=1e-2, rtol=2e-2)
    acc = torch.zeros(H, N + 7, dtype=torch.float32, device="cuda")
    ops.importance_accumulate_prefill(acc, qd, krd, N, cos, sin, P, lse, causal=True)
    torch.cuda.synchronize()
    mask = np.where(orc.causal_mask(B, ql, N, "f32") < 0, -np.inf, 0).astype(np.float32)
    want = orc.cascade_importance_accumulate(np.zeros((H, N), np.float32), host(stash), mask)     # from the kernel's own stash
    np.testing.assert_allclose(host(acc)[:, :N], want, rtol=1e-2, atol=2e-3)
    assert float(host(acc)[:, N:].max()) == 0.0
    np.testing.assert_allclose(host(acc)[:, :N].sum(1), np.full(H, B * ql), rtol=1e-3)        # every row's probabilities sum to 1
    # the row statistics are what the accumulation needs: softmax of the stash row == exp(s - m) / l
    pm = np.exp(host(stash)[0, 1, 5, :P + 6] - host(lse)[0, 1, 5, 0]) / host(lse)[0, 1, 5, 1]
    np.testing.assert_allclose(pm, orc.softmax_probs(host(stash)[0, 1, 5, :P + 6]), rtol=2e-3, atol=1e-6)
    # and the stash-based kernel agrees
    acc2 = torch.zeros(H, N, dtype=torch.float32, device="cuda")
    ops.importance_accumulate(acc2, stash, None, None, causal=True)
    np.testing.assert_allclose(host(acc)[:, :N], host(acc2), rtol=1e-2, atol=2e-3)


def test_cascade_prefill_8192_llama7b_without_the_stash():
    """configs[3] geometry (H = 32, q = N = 8192, bf16) in cascade mode: no 4 GiB stash; every query row's probabilities
    sum to 1, so each head's importance sums to the number of rows; spot columns against a direct evaluation."""
    from spatten_amd import ops
    dt, B, H, d, N = "bf16", 1, 32, 128, 8192
    tdt = TORCH_DT[dt]
    gen = torch.Generator(device="cuda").manual_seed(8)
    K = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    V = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    Q = torch.randn(B, H, N, d, device="cuda", generator=gen).to(tdt)
    cos, sin = ops.rope_table(N, d, tdt, "cuda")
    Kr = ops.rope_single(K, cos, sin)
    lse = torch.empty(B, H, N, 2, dtype=torch.float32, device="cuda")
    ops.attn_prefill(Q, Kr, V, N, cos, sin, 0, causal=True, lse=lse)
    acc = torch.zeros(H, N, dtype=torch.float32, device="cuda")
    ops.importance_accumulate_prefill(acc, Q, Kr, N, cos, sin, 0, lse, causal=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(acc).sum(1), np.full(H, N), rtol=2e-3)
    # column j of head h, directly: sum_i>=j softmax_i[j]  (torch fp32 on the device, two heads)
    Qr = ops.rope_single(Q, cos, sin)
    for h in (0, 17):
        sc = (Qr[0, h].float() @ Kr[0, h].float().T) / d ** 0.5
        sc = sc.masked_fill(torch.ones(N, N, dtype=torch.bool, device="cuda").triu(1), float("-inf"))
        want = torch.softmax(sc, dim=-1).sum(0)
        np.testing.assert_allclose(host(acc[h]), host(want), rtol=3e-2, atol=3e-3)


@pytest.mark.parametrize("dt,Hkv", [("bf16", 8), ("f16", 2), ("f32", 8)])
def test_local_value_pruning_vs_oracle(dt, Hkv):
    from spatten_amd.cascade import local_v_decode
    B, H, d, P = 2, 8, 128, 700
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 21)
    for keep in (N // 5, 64, N):
        out, st = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep)
        probs = orc.softmax_probs(host(st))                       # probabilities of the kernel's own stash
        want = orc.local_value_prune(probs, orc.repeat_kv(vc, H // Hkv), keep).reshape(B, H * d)
        np.testing.assert_allclose(host(out), orc.round_dt(want, dt), **OUT_TOL[dt])
        np.testing.assert_allclose(host(st), stash[:, :, 0] if stash.ndim == 4 else stash, rtol=2 ** -6, atol=1e-4)


def test_local_value_pruning_long_context_vs_oracle():
    """20000 rows: the pipelined scores-only pass, the 32-scores-per-thread select and a P.V gather split over ~20
    workgroups per head"""
    from spatten_amd.cascade import local_v_decode
    dt, B, H, d, P = "bf16", 1, 8, 128, 19999
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, H, d, P, dt, 23)
    for keep in (6000, 1500):
        out, st = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep)
        probs = orc.softmax_probs(host(st))
        want = orc.local_value_prune(probs, vc, keep).reshape(B, H * d)
        np.testing.assert_allclose(host(out), orc.round_dt(want, dt), **OUT_TOL[dt])
        np.testing.assert_allclose(host(st), stash[:, :, 0] if stash.ndim == 4 else stash, rtol=2 ** -6, atol=1e-4)
        out2, _ = local_v_decode(qd, krd, vd, N, cos, sin, N - 1, keep)          # workspace re-armed: same bits again
        assert torch.equal(out, out2)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_pv_gather_split_counts_and_mask_vs_torch(dt):
    """spatten_pv_gather over kept lists from 1 to 9000 rows (one slice, a few, the 64-slice cap), GQA, with a mask,
    repeated on one workspace — against the same sum in torch fp32"""
    from spatten_amd import ops
    tdt = TORCH_DT[dt]
    B, H, Hkv, d, N = 2, 8, 4, 128, 12000
    g = torch.Generator(device="cuda").manual_seed(3)
    stash = (torch.randn(B, H, N, device="cuda", generator=g) * 2).to(tdt)
    V = torch.randn(B, Hkv, N, d, device="cuda", generator=g).to(tdt)
    mask = torch.zeros(B, N, device="cuda", dtype=tdt)
    mask[:, ::7] = -3.0
    lg = (stash.float() + mask[:, None, :].float()).to(tdt).float()
    m = lg.max(-1).values
    lse = torch.stack([m, torch.exp(lg - m[..., None]).sum(-1)], -1).contiguous()
    for k in (1, 17, 64, 65, 700, 4097, 9000):
        idx = torch.stack([torch.randperm(N, device="cuda", generator=g)[:k].sort().values for _ in range(B * H)]).to(torch.int32)
        for rep in range(2):
            out = ops.pv_gather(stash, lse, V, idx, mask=mask)
        p = torch.exp(torch.gather(lg.reshape(B * H, N), 1, idx.long()) - m.reshape(B * H, 1)) / lse[..., 1].reshape(B * H, 1)
        Vh = V.repeat_interleave(H // Hkv, dim=1).reshape(B * H, N, d).float()
        want = torch.einsum("uk,ukd->ud", p, torch.gather(Vh, 1, idx.long()[..., None].expand(-1, -1, d))).reshape(B, H * d)
        tol = dict(rtol=2e-2, atol=2e-2) if dt == "bf16" else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out.float(), want, **tol)


def test_head_scores_and_pruned_decode():
    from spatten_amd import ops
    from spatten_amd.cascade import HeadPruner
    dt, B, H, d, P = "bf16", 2, 8, 128, 400
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, H, d, P, dt, 31)
    full = ops.attn_decode(qd, None, krd, vd, N, cos, sin, N - 1)
    hp = HeadPruner(H, "cuda")
    hp.observe(full[:, None, :])
    hp.observe(full[:, None, :])
    want = 2 * orc.head_scores(host(full)[:, None, :], H)
    np.testing.assert_allclose(host(hp.scores), want, rtol=1e-5)
    keep = hp.select(6)
    assert np.array_equal(keep.cpu().numpy(), orc.head_prune_select(want, 6))
    # pruned decode: only the kept heads run; their outputs are identical, the others stay untouched
    out = torch.full_like(full, float("nan"))
    st = torch.full((B, H, N), float("nan"), dtype=TORCH_DT[dt], device="cuda")
    ops.attn_decode(qd, None, krd, vd, N, cos, sin, N - 1, out=out, scores=st, head_ids=keep)
    torch.cuda.synchronize()
    o3, f3 = out.view(B, H, d), full.view(B, H, d)
    kept = keep.cpu().numpy().tolist()
    for h in range(H):
        if h in kept:
            assert torch.equal(o3[:, h], f3[:, h])
        else:
            assert torch.isnan(o3[:, h].float()).all() and torch.isnan(st[:, h].float()).all()


@pytest.mark.parametrize("dt,d,Hkv", [("bf16", 128, 8), ("f16", 64, 8), ("f32", 128, 4)])
def test_progressive_quant_planes_and_decode_vs_oracle(dt, d, Hkv):
    """MSB/LSB planes bit exact vs the oracle's quantiser; decode = MSB pass, LSB refetch below the threshold, P·V."""
    from spatten_amd import ops
    B, H, P = 2, 8, 500
    q, kc, vc, stash, (qd, krd, vd, cos, sin, N) = setup_decode(B, H, Hkv, d, P, dt, 41)
    planes = ops.PQPlanes(B, Hkv, N + 9, d, "cuda")
    ops.pq_pack(krd, planes, 0, N)
    kr_host = host(krd)
    msb, lsb, scale = orc.pq_quantize(kr_host)
    gm, gl, gs = planes.unpack(N)
    assert np.array_equal(gm, msb) and np.array_equal(gl, lsb) and np.array_equal(gs, scale)
    c, s = orc.rope_table(N, d, dt)
    qr = orc.apply_rotary_pos_emb_single(q, c, s, np.full((B, 1), N - 1), dt)[:, :, 0]
    rep = lambda a: orc.repeat_kv(a, H // Hkv)
    # thresholds: never refetch, always refetch, and a mix (median of the pass-1 max probabilities)
    k1 = orc.pq_dequant(rep(msb), None, rep(scale))
    p1max = orc.softmax_probs(np.einsum("bhd,bhld->bhl", qr, k1) / np.float32(np.sqrt(d))).max(-1)
    for thr in (0.0, 2.0, float(np.median(p1max))):
        want, need = orc.pq_decode_attention(qr, rep(msb), rep(lsb), rep(scale), rep(vc), thr)
        need_dev = torch.full((B * H,), -1, dtype=torch.int32, device="cuda")
        out = ops.attn_decode_pq(qd, planes, vd, N, cos, sin, N - 1, thr, need_lsb=need_dev)
        torch.cuda.synchronize()
        assert np.array_equal(need_dev.cpu().numpy().reshape(B, H).astype(bool), need), thr
        np.testing.assert_allclose(host(out).reshape(B, H, d), orc.round_dt(want, dt), **OUT_TOL[dt])
    # 8-bit keys are close to, but not the same as, the un-quantised attention: sanity bound
    full = ops.attn_decode(qd, None, krd, vd, N, cos, sin, N - 1)
    out8 = ops.attn_decode_pq(qd, planes, vd, N, cos, sin, N - 1, 2.0)
    assert float((full.float() - out8.float()).abs().max()) < 0.05


@pytest.mark.parametrize("with_acc", [False, True])
def test_layer_cascade_event_in_three_launches_equals_the_per_layer_composition(with_acc):
    """spatten_prune_layer_cascade (one workgroup per head walking the layers + ragged gathers) against the round-2
    composition of per-layer launches — cascade_rank -> topk_select -> kv_compact (+ shadow) -> id gather -> accumulator
    compaction — at Llama-like geometry with ragged layers (different lengths, shrinking keeps, ids known from an earlier
    prune plus rows appended since): kept positions, new ids, K / V / shadow rows and accumulators bit for bit."""
    from spatten_amd import ops
    tdt, B, H, d, start, recent, coming = torch.bfloat16, 1, 32, 128, 4, 256, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    lens = [1500, 1400, 1400, 1300, 1210]
    keeps = [700, 650, 650, 500, 400]
    nl = len(lens)
    appended = 90                                             # rows appended to every layer since the last prune
    cos, sin = ops.rope_table(2048, d, tdt, "cuda")
    Ks = [torch.randn(B, H, L + 13, d, device="cuda", generator=g).to(tdt) for L in lens]
    Vs = [torch.randn(B, H, L + 13, d, device="cuda", generator=g).to(tdt) for L in lens]
    his = [min(L - recent + coming, L) for L in lens]
    # ids: ascending per head; the slots of layer l hold a subset of layer l-1's tokens (what an earlier cascade leaves)
    pool = torch.arange(5000, device="cuda")
    known, prev = [], None
    for l, L in enumerate(lens):
        nk = L - appended
        rows = []
        for h in range(H):
            src = pool[:4000] if prev is None else prev[h]
            pick = torch.randperm(src.numel(), device="cuda", generator=g)[:nk].sort().values
            rows.append(src[pick])
        ids = torch.stack(rows).to(torch.int32).contiguous()
        known.append(ids)
        prev = [r for r in ids.long()]
    id_base = 6000
    if with_acc:
        scores = [torch.rand(H, L + 50, device="cuda", generator=g) for L in lens]          # fp32 accumulators, wider than the cache
        accs = scores
    else:
        scores = [torch.randn(H, L, device="cuda", generator=g).to(tdt) for L in lens]
        accs = None
    caps = [(start + k + (L - hi) + coming + 127) // 128 * 128 for L, hi, k in zip(lens, his, keeps)]
    Kn, Vn, Krn, idxs, new_ids, new_accs = ops.prune_layer_cascade(
        [s[:, :L] for s, L in zip(scores, lens)], known, id_base, [K[:, :, :L] for K, L in zip(Ks, lens)],
        [V[:, :, :L] for V, L in zip(Vs, lens)], lens, his, keeps, start, caps, (cos, sin),
        None if accs is None else accs)
    torch.cuda.synchronize()
    # ---- the same event, one layer at a time
    prev_ids = None
    for l, L in enumerate(lens):
        fresh = (torch.arange(appended, dtype=torch.int32, device="cuda") + id_base)[None].expand(H, appended)
        ids = torch.cat([known[l], fresh], 1).contiguous()
        sc = scores[l][:, :L].contiguous()
        rank = sc if prev_ids is None else ops.cascade_rank(sc, ids, prev_ids)
        idx = ops.topk_select(rank, start, his[l], keeps[l])
        k, v, kr = ops.kv_compact(Ks[l][:, :, :L], Vs[l][:, :, :L], idx, start, his[l], L=L, capacity=caps[l], rope=(cos, sin))
        prev_ids = ops.gather_rows_i32(ids, idx, start, his[l], L)
        assert torch.equal(idxs[l], idx), l
        assert torch.equal(new_ids[l], prev_ids), l
        assert torch.equal(Kn[l], k) and torch.equal(Vn[l], v) and torch.equal(Krn[l], kr), l
        if with_acc:
            want = ops.importance_compact(accs[l], idx, start, his[l], L, new_accs[l].shape[1])
            n = k.shape[2]
            assert torch.equal(new_accs[l][:, :n], want[:, :n]) and (new_accs[l][:, n:] == 0).all(), l
    # a window that cannot supply layer_keep candidates is refused (reference: torch.topk RuntimeError)
    with pytest.raises(ValueError):
        ops.prune_layer_cascade([s[:, :L] for s, L in zip(scores, lens)], known, id_base, [K[:, :, :L] for K, L in zip(Ks, lens)],
                                [V[:, :, :L] for V, L in zip(Vs, lens)], lens, [start + 10] * nl, keeps, start, caps, (cos, sin))


def test_layer_cascade_event_replays_from_a_captured_graph_and_from_its_plan():
    """ops.LayerCascadePlan: the host preparation once, run() = the C call alone.  The event forks its gathers onto the library's
    side stream and joins it again (include/spatten.h: spatten_prune_layer_cascade) — a stream capture follows both, so the whole
    event replays from a HIP graph; results equal the one-shot op bit for bit, from the plan, from the graph, and with the legs
    switched (4 legs over 6 layers vs the per-call default)."""
    from spatten_amd import ops
    tdt, B, H, d, start, recent = torch.bfloat16, 1, 8, 128, 4, 64
    g = torch.Generator(device="cuda").manual_seed(9)
    lens = [900, 900, 880, 880, 860, 860]
    keeps = [400, 380, 360, 340, 320, 300]
    nl = len(lens)
    his = [L - recent for L in lens]
    cos, sin = ops.rope_table(1024, d, tdt, "cuda")
    Ks = [torch.randn(B, H, L, d, device="cuda", generator=g).to(tdt) for L in lens]
    Vs = [torch.randn(B, H, L, d, device="cuda", generator=g).to(tdt) for L in lens]
    scores = [torch.randn(H, L, device="cuda", generator=g).to(tdt) for L in lens]
    known = [torch.arange(L - 50, device="cuda", dtype=torch.int32)[None].expand(H, L - 50).contiguous() for L in lens]
    caps = [512] * nl
    want = ops.prune_layer_cascade(scores, known, 5000, Ks, Vs, lens, his, keeps, start, caps, (cos, sin))
    torch.cuda.synchronize()
    dst = ([torch.zeros(B, H, 512, d, dtype=tdt, device="cuda") for _ in range(nl)],
           [torch.zeros(B, H, 512, d, dtype=tdt, device="cuda") for _ in range(nl)],
           [torch.zeros(B, H, 512, d, dtype=tdt, device="cuda") for _ in range(nl)])
    plan = ops.LayerCascadePlan(scores, known, 5000, Ks, Vs, lens, his, keeps, start, caps, (cos, sin), dst=dst)

    def check(got):
        torch.cuda.synchronize()
        for l in range(nl):
            assert torch.equal(got[3][l], want[3][l]) and torch.equal(got[4][l], want[4][l]), l
            assert torch.equal(got[0][l], want[0][l]) and torch.equal(got[1][l], want[1][l]) and torch.equal(got[2][l], want[2][l]), l
    check(plan.run())                       # (also creates the side stream outside any capture)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        plan.run()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            plan.run()
    for t in dst[0] + dst[1] + dst[2]:
        t.zero_()
    for t in plan.result[3] + plan.result[4]:
        t.zero_()
    torch.cuda.synchronize()
    graph.replay()
    check(plan.result)
    graph.replay()                          # a second replay: the fork events are re-recorded inside the graph
    check(plan.result)


def test_cascade_prune_of_a_grouped_query_cache_vs_oracle(capsys):
    """importance_mode="cascade" on a grouped-query cache (Hkv < H): a key's importance is the sum of its group's
    accumulator rows (the reference-mode rule for GQA), the kept rows are the oracle's top-k of that sum per KV head, and
    every QUERY head's accumulator row follows its KV head's row map."""
    from spatten_amd.extensions import SpattenExtensions
    from spatten_amd.kv_cache_token_pruning import SpAttenKVCache
    B, H, Hkv, d, L, layers, coming = 1, 8, 2, 64, 300, 3, 10
    G = H // Hkv
    cache = SpAttenKVCache(start_size=4, recent_size=40, important_size=50, importance_mode="cascade")
    cache.ext = SpattenExtensions(cache, layers, cascade=True)
    rng = np.random.default_rng(5)
    past, accs = [], []
    for i in range(layers):
        K = orc.round_dt(rng.standard_normal((B, Hkv, L, d)).astype(np.float32), "bf16")
        V = orc.round_dt(rng.standard_normal((B, Hkv, L, d)).astype(np.float32), "bf16")
        acc = rng.random((H, L + 16)).astype(np.float32)
        past.append([dev(K, "bf16"), dev(V, "bf16")])
        accs.append(acc)
        cache.ext.layers[i].acc = torch.from_numpy(acc).cuda()
    new = cache.apply_token_pruning(past, coming, [None] * layers)
    lo, hi = 4, L - 40 + coming
    for i in range(layers):
        score = accs[i][:, :L].reshape(Hkv, G, L).sum(1)
        idx = orc.topk_window(score, lo, hi, 50)
        Kw, Vw = orc.kv_compact(host(past[i][0]), host(past[i][1]), idx, 4, hi)
        assert np.array_equal(host(new[i][0]), Kw) and np.array_equal(host(new[i][1]), Vw)
        want = np.stack([np.concatenate([accs[i][h, :4], accs[i][h, idx[h // G]], accs[i][h, hi:L]]) for h in range(H)])
        got = cache.ext.layers[i].acc[:, :want.shape[1]].cpu().numpy()
        assert np.array_equal(got, want)
