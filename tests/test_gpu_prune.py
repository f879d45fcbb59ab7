"""Prune event (HIP, through the C ABI) vs the reference goldens and the oracle: bit exact.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import TORCH_DT, dev, golden, host

pytestmark = pytest.mark.gpu


def prune_inputs(H, L, d, qs, dt, seed, bump):
    stash = orc.synth_normal(seed + 1000 * bump, 5, (1, H, qs, L), dt)
    K = orc.synth_normal(seed, 6, (1, H, L, d), dt)
    V = orc.synth_normal(seed, 7, (1, H, L, d), dt)
    return stash, K, V


def test_prune_matches_reference_goldens_bit_exact(capsys):
    from spatten_amd import SpAttenKVCache
    g = golden("g1_prune.npz")
    for m in g["meta"]:
        name, H, L, d, start, recent, important, c, qs, dt, seed, bump = m.split("|")
        H, L, d, start, recent, important, c, qs, seed, bump = map(int, (H, L, d, start, recent, important, c, qs, seed, bump))
        stash, K, V = prune_inputs(H, L, d, qs, dt, seed, bump)
        cache = SpAttenKVCache(start_size=start, recent_size=recent, important_size=important)
        past = [(dev(K, dt), dev(V, dt))]
        out = cache.apply_token_pruning(past, c, [dev(stash, dt)])
        torch.cuda.synchronize()
        assert isinstance(out, list) and isinstance(out[0], list)
        assert np.array_equal(host(out[0][0]), g[f"{name}_K"]), name
        assert np.array_equal(host(out[0][1]), g[f"{name}_V"]), name
        imp = host(cache.importance_score[0])
        if qs == 1:
            assert np.array_equal(imp, g[f"{name}_imp"]), name
        else:
            np.testing.assert_allclose(imp, g[f"{name}_imp"], rtol=1e-6, atol=1e-6)
        # inputs untouched
        assert np.array_equal(host(past[0][0]), K)
    assert "SpAttenKVCache: keep start" in capsys.readouterr().out      # the reference prints this banner


def test_prune_c2_scale_indices_bit_exact():
    from spatten_amd import ops
    g = golden("g2_prune_c2.npz")
    for tag in ("c0", "c64"):
        H, L, start, recent, important, c, seed, bump = map(int, g[f"{tag}_meta"])
        stash = orc.synth_normal(seed + 1000 * bump, 5, (1, H, 1, L), "f32")
        idx = ops.topk_select(dev(stash[0, :, 0], "f32"), start, L - recent + c, important)
        kept = g[f"{tag}_kept"].astype(np.int64)
        assert np.array_equal(idx.cpu().numpy(), kept[:, start:start + important])


def test_passthrough_and_none():
    from spatten_amd import SpAttenKVCache
    cache = SpAttenKVCache(start_size=4, recent_size=32, important_size=28)
    assert cache.apply_token_pruning(None, 5, []) is None
    past = [(torch.zeros(1, 2, 40, 8, device="cuda"), torch.zeros(1, 2, 40, 8, device="cuda"))]
    assert cache.apply_token_pruning(past, 24, [torch.zeros(1, 2, 1, 40, device="cuda")]) is past
    with pytest.raises(ValueError):      # window too short: num_coming > recent, short cache
        SpAttenKVCache(4, 8, 50).apply_token_pruning(past, 30, [torch.zeros(1, 2, 1, 40, device="cuda")])


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_topk_ties_nan_inf_vs_oracle(dt):
    from spatten_amd import ops
    rng = np.random.default_rng(3)
    H, L = 6, 1000
    s = orc.round_dt(rng.standard_normal((H, L)).astype(np.float32), dt)
    s[0] = 0.0                                    # all tied
    s[1] = np.round(s[1] * 2) / 2                  # heavy duplicates
    s[2, ::7] = np.inf
    s[2, 5::11] = -np.inf
    s[3, 3::13] = np.nan                           # NaN ranks largest (torch.topk)
    s[4, :] = orc.round_dt(np.where(rng.random(L) < 0.5, 0.0, -0.0).astype(np.float32), dt)   # +-0 tie
    for lo, hi, k in ((0, L, 1), (0, L, L), (4, 900, 300), (10, 11, 1), (3, 997, 994), (100, 612, 256), (0, L, 40)):
        want = orc.topk_window(s, lo, hi, k)
        # the output buffer is over-allocated and poisoned: a kernel that keeps more than k (e.g. k <= #NaN) is caught
        from spatten_amd import _lib
        idx = torch.full((H, k + 64), -7, dtype=torch.int32, device="cuda")
        sd = dev(s, dt)
        rc = _lib.load().spatten_topk_select({"f32": 0, "f16": 1, "bf16": 2}[dt], sd.data_ptr(), sd.stride(0), H, lo, hi, k,
                                             idx.data_ptr(), idx.stride(0), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        got = idx.cpu().numpy()
        assert np.array_equal(got[:, :k], want), (dt, lo, hi, k)
        assert (got[:, k:] == -7).all(), (dt, lo, hi, k)


def test_topk_large_window_and_bf16_threshold_ties():
    from spatten_amd import ops
    # bf16 random rows at N=4096 tie at the threshold in ~6% of rows: the contract there is
    # all > thr kept + the lowest-index == thr, which the oracle restates
    s = orc.synth_normal(9, 0, (32, 4096), "bf16")
    want = orc.topk_window(s, 4, 3072, 1020)
    assert np.array_equal(ops.topk_select(dev(s, "bf16"), 4, 3072, 1020).cpu().numpy(), want)
    s = orc.synth_normal(10, 0, (3, 100000), "f32")
    for k in (1, 777, 50000, 99990):
        assert np.array_equal(ops.topk_select(dev(s, "f32"), 5, 99995, k).cpu().numpy(), orc.topk_window(s, 5, 99995, k))
    with pytest.raises(ValueError):
        ops.topk_select(dev(s, "f32"), 5, 10, 6)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_topk_register_resident_windows_vs_oracle(dt):
    """windows of 4k..32k scores (the 16- and 32-scores-per-thread instantiations) and one beyond (the histogram
    kernel), with ties at the threshold, infinities and NaN"""
    from spatten_amd import ops
    rng = np.random.default_rng(11)
    H, L = 5, 40000
    s = orc.round_dt(rng.standard_normal((H, L)).astype(np.float32), dt)
    s[1] = np.round(s[1] * 4) / 4                  # heavy duplicates: the threshold value repeats ~thousands of times
    s[2, ::9] = np.inf
    s[2, 4::17] = -np.inf
    s[3, 3::1001] = np.nan
    s[4] = 0.0
    sd = dev(s, dt)
    for lo, hi, k in ((0, 16384, 4915), (5, 12005, 1), (7, 12007, 12000), (0, 20000, 6000), (100, 32868, 32768),
                      (100, 32868, 9000), (3, 5003, 2500), (0, 40000, 20000)):
        want = orc.topk_window(s, lo, hi, k)
        assert np.array_equal(ops.topk_select(sd, lo, hi, k).cpu().numpy(), want), (dt, lo, hi, k)


@pytest.mark.parametrize("rows", [3, 130])
def test_topk_every_dispatch_boundary_vs_oracle(rows):
    """window widths on both sides of every instantiation boundary of the select dispatch (1024 / 4096 / 8192 / 16384 /
    32768 scores; few windows -> 16-wave workgroups, many windows -> 4-wave ones), random k, bf16 scores with ties"""
    from spatten_amd import ops
    rng = np.random.default_rng(100 + rows)
    L = 33000
    s = orc.round_dt(np.round(rng.standard_normal((rows, L)).astype(np.float32) * 8) / 8, "bf16")
    sd = dev(s, "bf16")
    for W in (1, 2, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 8193, 16384, 16385, 32767, 32768, 32769):
        lo = int(rng.integers(0, L - W + 1))
        for k in sorted({1, W, int(rng.integers(1, W + 1)), max(1, W // 2)}):
            got = ops.topk_select(sd, lo, lo + W, k).cpu().numpy()
            assert np.array_equal(got, orc.topk_window(s, lo, lo + W, k)), (rows, W, lo, k)


@pytest.mark.parametrize("dt,d", [("bf16", 128), ("f32", 64), ("f16", 80), ("f32", 8)])
def test_kv_compact_batch_heads_vs_oracle(dt, d):
    from spatten_amd import ops
    B, H, L = 2, 5, 700
    K = orc.synth_normal(21, 0, (B, H, L, d), dt)
    V = orc.synth_normal(21, 1, (B, H, L, d), dt)
    s = orc.synth_normal(21, 2, (H, L), "f32")
    for start, tail_lo, k in ((4, 500, 300), (0, 700, 128), (7, 650, 1), (4, 100, 96)):
        idx = orc.topk_window(s, start, tail_lo, k)
        wk, wv = orc.kv_compact(K, V, idx, start, tail_lo)
        rope = None
        if d % 16 == 0:      # also rebuild the rotated shadow of the new cache in the same pass
            c, sn = orc.rope_table(900, d, dt)
            rope = (dev(c[:, : d // 2], dt), dev(sn[:, : d // 2], dt))
        gk, gv, gkr = ops.kv_compact(dev(K, dt), dev(V, dt), torch.from_numpy(idx).cuda(), start, tail_lo,
                                     capacity=900, rope=rope)
        assert gk.shape == wk.shape
        assert np.array_equal(host(gk), wk) and np.array_equal(host(gv), wv)
        if rope is not None:   # shadow row r = RoPE(new K row r, position r)  (modify_llama.py:103-104), bit exact
            want = orc.apply_rotary_pos_emb_single(wk, c, sn, np.arange(wk.shape[2])[None], dt)
            assert np.array_equal(host(gkr), want)


def test_h1_and_batch2_supported():
    """The reference raises IndexError for H == 1 and B == 2; the HIP path handles both."""
    from spatten_amd import SpAttenKVCache
    for B, H in ((1, 1), (2, 3)):
        L, d = 200, 16
        K = orc.synth_normal(3, 0, (B, H, L, d), "f32")
        V = orc.synth_normal(3, 1, (B, H, L, d), "f32")
        stash = orc.synth_normal(3, 2, (B, H, 1, L), "f32")
        cache = SpAttenKVCache(4, 40, 30)
        out = cache.apply_token_pruning([(dev(K, "f32"), dev(V, "f32"))], 10, [dev(stash, "f32")])
        want, _ = orc.apply_token_pruning([(K, V)], 10, [stash], 4, 40, 30, "f32")
        assert np.array_equal(host(out[0][0]), want[0][0]) and np.array_equal(host(out[0][1]), want[0][1])


def test_protocol_trajectory_on_gpu():
    """G5: multi-turn caller protocol (run_spatten_llama.py:60-87): L trajectory, kept rows, passthrough identity."""
    from spatten_amd import SpAttenKVCache
    g = golden("g5_protocol.npz")
    H, d, start, recent, important = map(int, g["params"])
    cache = SpAttenKVCache(start, recent, important)
    past, row = None, 0
    for turn, (plen, gen) in enumerate(g["turns"]):
        if past is not None:
            Lp = past[0][0].shape[2]
            stash = orc.synth_normal(50 + turn, 5, (1, H, 1, Lp), "f32")
            new = cache.apply_token_pruning(past, int(plen) + 20, [dev(stash, "f32")])
            t = g["traj"][row]
            assert (turn, Lp, int(plen) + 20, new[0][0].shape[2], int(new is past)) == tuple(int(x) for x in t)
            assert np.array_equal(host(new[0][0])[0, :, :, 0].astype(np.int64), g[f"kept_{row}"])
            past, row = new, row + 1
        n_new = int(plen + gen)
        ids = (1000 * (turn + 1) + np.arange(n_new)).astype(np.float32)
        new_rows = dev(np.broadcast_to(ids[None, None, :, None], (1, H, n_new, d)), "f32")
        past = [(new_rows, new_rows)] if past is None else \
            [(torch.cat([past[0][0], new_rows], 2), torch.cat([past[0][1], new_rows], 2))]
    assert row == len(g["traj"])
    assert cache.n_pruned_total == int((g["traj"][:, 1] - g["traj"][:, 3]).sum())


def test_prune_c2_full_size_all_layers_properties():
    """Llama-2-7B geometry (H=32, L=4096, d=128, bf16), 4 layers in one batched launch: bit-exact vs the oracle
    on layer 0 and size-independent properties on all (ascending, window-bounded, start/tail preserved)."""
    from spatten_amd import SpAttenKVCache
    H, L, d, nl, dt = 32, 4096, 128, 4, "bf16"
    cache = SpAttenKVCache(4, 1024, 1020)
    gen = torch.Generator(device="cuda").manual_seed(0)
    past = [(torch.randn(1, H, L, d, device="cuda", generator=gen).to(torch.bfloat16),
             torch.randn(1, H, L, d, device="cuda", generator=gen).to(torch.bfloat16)) for _ in range(nl)]
    stash = [torch.randn(1, H, 1, L, device="cuda", generator=gen).to(torch.bfloat16) for _ in range(nl)]
    for c in (0, 64):
        out = cache.apply_token_pruning(past, c, stash)
        torch.cuda.synchronize()
        Lp = 4 + 1020 + (1024 - c)
        idx = cache.keep_indices.cpu().numpy()
        assert idx.shape == (nl, H, 1020)
        assert np.all(np.diff(idx, axis=-1) > 0) and idx.min() >= 4 and idx.max() < L - 1024 + c
        for l in range(nl):
            K, V = past[l]
            Kn, Vn = out[l]
            assert Kn.shape == (1, H, Lp, d)
            assert torch.equal(Kn[:, :, :4], K[:, :, :4]) and torch.equal(Vn[:, :, Lp - (1024 - c):], V[:, :, L - 1024 + c:])
            ii = torch.from_numpy(idx[l]).cuda().long()
            assert torch.equal(Kn[0, :, 4:1024], torch.gather(K[0], 1, ii[:, :, None].expand(-1, -1, d)))
            assert torch.equal(Vn[0, :, 4:1024], torch.gather(V[0], 1, ii[:, :, None].expand(-1, -1, d)))
        want = orc.topk_window(host(stash[0])[0, :, 0], 4, L - 1024 + c, 1020)
        assert np.array_equal(idx[0], want)


def test_grouped_query_cache_is_pruned_by_the_sum_of_its_group():
    """GQA (H query heads on Hkv cached heads): the reference cannot prune such a cache (its [H, L] mask meets a
    [Hkv, L, d] tensor); here a cached key's importance is the sum of its group's stash rows."""
    from spatten_amd import SpAttenKVCache
    H, Hkv, L, d, dt = 8, 2, 300, 64, "f32"
    stash = orc.synth_normal(21, 5, (1, H, 1, L), dt)
    K = orc.synth_normal(21, 6, (1, Hkv, L, d), dt)
    V = orc.synth_normal(21, 7, (1, Hkv, L, d), dt)
    cache = SpAttenKVCache(start_size=4, recent_size=64, important_size=50)
    out = cache.apply_token_pruning([(dev(K, dt), dev(V, dt))], 20, [dev(stash, dt)])
    score = torch.from_numpy(stash[0, :, 0]).reshape(Hkv, H // Hkv, L).sum(1).numpy()
    idx = orc.topk_window(score, 4, L - 64 + 20, 50)
    wk, wv = orc.kv_compact(K, V, idx, 4, L - 64 + 20)
    assert np.array_equal(cache.keep_indices[0].cpu().numpy(), idx)
    assert np.array_equal(host(out[0][0]), wk) and np.array_equal(host(out[0][1]), wv)
    assert cache.importance_score[0].shape == (Hkv, L)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_global_token_scope_keeps_one_set_per_layer_vs_oracle(dt):
    """token_scope="global" (README.md:21, workloads/small.csv:1; parity unpinned): the kept set is ranked by the importance
    summed over the heads and shared by all of them — K', V' and the indices equal the oracle's, bit for bit; also when the
    heads are split over two head-parallel ranks (the [layers, L] sums all-reduced once per event)."""
    from spatten_amd import SpAttenKVCache
    from spatten_amd.parallel import HeadParallel
    H, L, d, layers, start, recent, important, c = 8, 600, 64, 3, 4, 100, 200, 16
    past_np, stash_np = [], []
    for l in range(layers):
        stash, K, V = prune_inputs(H, L, d, 1, dt, 41 + l, 0)
        past_np.append((K, V)); stash_np.append(stash)
    scores = [orc.importance(s, dt) for s in stash_np]
    want, want_idx = orc.global_token_prune(past_np, c, scores, start, recent, important)
    cache = SpAttenKVCache(start_size=start, recent_size=recent, important_size=important, token_scope="global")
    out = cache.apply_token_pruning([(dev(K, dt), dev(V, dt)) for K, V in past_np], c, [dev(s, dt) for s in stash_np])
    torch.cuda.synchronize()
    for l in range(layers):
        idx = cache.keep_indices[l].cpu().numpy()
        assert np.array_equal(idx, want_idx[l])
        assert all(np.array_equal(idx[0], idx[h]) for h in range(H))          # one set for every head
        assert np.array_equal(host(out[l][0]), want[l][0]) and np.array_equal(host(out[l][1]), want[l][1])
    # two head-parallel ranks, run one after the other: each sums its own heads, the "all-reduce" adds the other's
    halves = []
    for rank in range(2):
        hs = slice(rank * H // 2, (rank + 1) * H // 2)
        other = slice((1 - rank) * H // 2, (2 - rank) * H // 2)
        far = torch.stack([dev(sc[other], dt).sum(0, dtype=torch.float64) for sc in scores])
        hp = HeadParallel(H, rank=rank, world=2, gather_fn=lambda t, r: t, reduce_fn=lambda t, r, far=far: t + far)
        cr = SpAttenKVCache(start_size=start, recent_size=recent, important_size=important, token_scope="global")
        cr.head_parallel = hp
        o = cr.apply_token_pruning([(dev(K[:, hs], dt), dev(V[:, hs], dt)) for K, V in past_np], c,
                                   [dev(s[:, hs], dt) for s in stash_np])
        halves.append(o)
    torch.cuda.synchronize()
    for l in range(layers):
        got_k = np.concatenate([host(halves[0][l][0]), host(halves[1][l][0])], axis=1)
        assert np.array_equal(got_k, want[l][0])


def test_token_scope_is_validated():
    from spatten_amd import SpAttenKVCache
    with pytest.raises(ValueError):
        SpAttenKVCache(token_scope="layer")
