"""Pin the numpy oracle (oracle/spatten_oracle.py) against the golden vectors captured
from the imported reference (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import spatten_oracle as orc

# Tolerances (stated): fp32 attention outputs atol/rtol 1e-5 (accumulation order differs between
# numpy and torch BLAS); 16-bit: one rounding step of the model dtype may flip where the fp32
# accumulations differ -> allow 2 ulp of the dtype on O(1) values.
TOL = {"f32": dict(atol=2e-5, rtol=1e-5), "bf16": dict(atol=2e-2, rtol=2e-2), "f16": dict(atol=3e-3, rtol=3e-3)}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _cksum(*arrs):
    return sum(float(np.asarray(a, dtype=np.float64).sum()) for a in arrs)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_rope_table_and_single(golden_dir, dt):
    g = _load(golden_dir, "g4_rope.npz")
    cos, sin = orc.rope_table(200, 64, dt)
    # table: torch.cos vs numpy cos may differ by 1 fp32 ulp; exact after 16-bit rounding
    tol = 0 if dt != "f32" else 2e-7
    assert np.abs(cos - g[f"rope_{dt}_cos"]).max() <= tol
    assert np.abs(sin - g[f"rope_{dt}_sin"]).max() <= tol
    cos128, sin128 = orc.rope_table(4096, 128, dt)
    rows = g[f"rope_{dt}_rows128"]
    tol128 = 2e-7 if dt == "f32" else 0
    assert np.abs(cos128[rows] - g[f"rope_{dt}_cos128"]).max() <= tol128
    assert np.abs(sin128[rows] - g[f"rope_{dt}_sin128"]).max() <= tol128
    x = orc.synth_normal(40, 0, (2, 4, 33, 64), dt)
    assert abs(_cksum(x) - float(g[f"rope_{dt}_inck"])) < 1e-9
    # use the reference's own table so the op itself is checked bit-exactly
    y = orc.apply_rotary_pos_emb_single(x, g[f"rope_{dt}_cos"], g[f"rope_{dt}_sin"], g[f"rope_{dt}_pos"], dt)
    assert np.array_equal(y, g[f"rope_{dt}_y"])


def _attn_cases(golden_dir):
    g = _load(golden_dir, "g3_attention.npz")
    return g, [m.split("|") for m in g["meta"]]


def attn_inputs(name, B, H, Hkv, d, P, ql, dt, seed):
    q = orc.synth_normal(seed, 0, (B, H, ql, d), dt)
    k = orc.synth_normal(seed, 1, (B, Hkv, ql, d), dt)
    v = orc.synth_normal(seed, 2, (B, Hkv, ql, d), dt)
    past = None
    if P > 0:
        past = (orc.synth_normal(seed, 3, (B, Hkv, P, d), dt), orc.synth_normal(seed, 4, (B, Hkv, P, d), dt))
    return q, k, v, past


def test_attention_core_matches_reference(golden_dir):
    g, metas = _attn_cases(golden_dir)
    assert len(metas) >= 15
    for name, B, H, Hkv, d, P, ql, mask_kind, dt, seed in metas:
        B, H, Hkv, d, P, ql, seed = map(int, (B, H, Hkv, d, P, ql, seed))
        q, k, v, past = attn_inputs(name, B, H, Hkv, d, P, ql, dt, seed)
        inck = _cksum(q, k, v) + (0.0 if past is None else _cksum(*past))
        assert abs(inck - float(g[f"{name}_inck"])) < 1e-6, name
        N = P + ql
        pos = np.tile(np.arange(P, N)[None], (B, 1))
        mask = None
        if mask_kind == "zeros":
            mask = np.zeros((B, 1, ql, N), np.float32)
        elif mask_kind == "causal":
            mask = orc.causal_mask(B, ql, N, dt)
        o, stash, (kc, vc) = orc.attention_core(q, k, v, None if past is None else past[0],
                                                None if past is None else past[1], pos, mask, dt)
        np.testing.assert_allclose(stash, g[f"{name}_stash"], err_msg=name, **TOL[dt])
        np.testing.assert_allclose(o, g[f"{name}_out"], err_msg=name, **TOL[dt])
        # the returned cache is the un-rotated concat (bit exact)
        assert abs(_cksum(kc) - float(g[f"{name}_kck"])) < 1e-6, name
        assert abs(_cksum(vc) - float(g[f"{name}_vck"])) < 1e-6, name
        if dt != "f32":
            # 16-bit stash: at most a tiny fraction of entries may differ, and then by one ulp
            frac = np.mean(stash != g[f"{name}_stash"])
            assert frac < 0.02, (name, frac)


def prune_inputs(H, L, d, qs, dt, seed, bump):
    stash = orc.synth_normal(seed + 1000 * bump, 5, (1, H, qs, L), dt)
    K = orc.synth_normal(seed, 6, (1, H, L, d), dt)
    V = orc.synth_normal(seed, 7, (1, H, L, d), dt)
    return stash, K, V


def test_prune_matches_reference_bit_exact(golden_dir):
    g = _load(golden_dir, "g1_prune.npz")
    for m in g["meta"]:
        name, H, L, d, start, recent, important, c, qs, dt, seed, bump = m.split("|")
        H, L, d, start, recent, important, c, qs, seed, bump = map(int, (H, L, d, start, recent, important, c, qs, seed, bump))
        stash, K, V = prune_inputs(H, L, d, qs, dt, seed, bump)
        imp = orc.importance(stash, dt)
        if qs == 1:
            assert np.array_equal(imp, g[f"{name}_imp"]), name
        else:
            np.testing.assert_allclose(imp, g[f"{name}_imp"], rtol=1e-6, atol=1e-6, err_msg=name)
        new_past, idxs = orc.apply_token_pruning([(K, V)], c, [stash], start, recent, important, dt)
        assert np.array_equal(new_past[0][0], g[f"{name}_K"]), name
        assert np.array_equal(new_past[0][1], g[f"{name}_V"]), name
        Lp = start + important + max(0, recent - c)
        assert new_past[0][0].shape == (1, H, Lp, d)
        assert np.all(np.diff(idxs[0], axis=1) > 0)


def test_prune_c2_scale_indices(golden_dir):
    g = _load(golden_dir, "g2_prune_c2.npz")
    for tag in ("c0", "c64"):
        H, L, start, recent, important, c, seed, bump = map(int, g[f"{tag}_meta"])
        stash = orc.synth_normal(seed + 1000 * bump, 5, (1, H, 1, L), "f32")
        score = orc.importance(stash, "f32")
        idx = orc.topk_window(score, start, L - recent + c, important)
        kept = g[f"{tag}_kept"].astype(np.int64)
        assert kept.shape == (H, start + important + max(0, recent - c))
        assert np.array_equal(kept[:, :start], np.tile(np.arange(start), (H, 1)))
        assert np.array_equal(kept[:, start:start + important], idx)
        assert np.array_equal(kept[:, start + important:], np.tile(np.arange(L - recent + c, L), (H, 1)))


def test_protocol_trajectory(golden_dir):
    g = _load(golden_dir, "g5_protocol.npz")
    H, d, start, recent, important = map(int, g["params"])
    turns = g["turns"]
    past = None
    row = 0
    assert orc.apply_token_pruning(None, 10, [], start, recent, important, "f32") == (None, None)
    for turn, (plen, gen) in enumerate(turns):
        if past is not None:
            Lp = past[0][0].shape[2]
            stash = orc.synth_normal(50 + turn, 5, (1, H, 1, Lp), "f32")
            new, _ = orc.apply_token_pruning(past, int(plen) + 20, [stash], start, recent, important, "f32")
            t = g["traj"][row]
            assert (turn, Lp, int(plen) + 20, new[0][0].shape[2], int(new is past)) == tuple(int(x) for x in t)
            assert np.array_equal(new[0][0][0, :, :, 0].astype(np.int64), g[f"kept_{row}"])
            past = new
            row += 1
        n_new = int(plen + gen)
        ids = (1000 * (turn + 1) + np.arange(n_new)).astype(np.float32)
        new_rows = np.broadcast_to(ids[None, None, :, None], (1, H, n_new, d)).copy()
        if past is None:
            past = [(new_rows, new_rows)]
        else:
            past = [(np.concatenate([past[0][0], new_rows], 2), np.concatenate([past[0][1], new_rows], 2))]
    assert row == len(g["traj"])


def test_topk_tie_policy_lowest_index_first():
    s = np.array([[1, 3, 3, 3, 2, 3, 0, 3]], np.float32)
    assert orc.topk_window(s, 0, 8, 2).tolist() == [[1, 2]]
    assert orc.topk_window(np.zeros((1, 300), np.float32), 0, 300, 5).tolist() == [[0, 1, 2, 3, 4]]
    s = np.array([[0.0, -0.0, np.nan, 1.0, -np.inf, np.inf]], np.float32)
    assert orc.topk_window(s, 0, 6, 3).tolist() == [[2, 3, 5]]      # NaN ranks largest (torch.topk)
    with pytest.raises(ValueError):
        orc.topk_window(np.zeros((1, 10), np.float32), 4, 8, 5)


def _bf16(u16):
    return (np.asarray(u16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def tie_rule_equal(kept_a, kept_b, score, lo, k):
    """Two selections of the k largest of a head's window agree under the tie rule: identical above the threshold value,
    and the same NUMBER of threshold-valued positions (which of those is unspecified in torch.topk)."""
    a, b = np.asarray(kept_a, np.int64), np.asarray(kept_b, np.int64)
    thr = min(score[a].min(), score[b].min())
    if score[a].min() != score[b].min():
        return False
    above_a, above_b = a[score[a] > thr], b[score[b] > thr]
    return np.array_equal(np.sort(above_a), np.sort(above_b)) and (score[a] == thr).sum() == (score[b] == thr).sum()


@pytest.mark.parametrize("name", ["c2", "c5"])
def test_fullsize_decode_goldens_of_the_reference(golden_dir, name):
    """G6: BASELINE.json configs[1] / [4] geometry through the reference forward itself (Llama-2-7B H = 32 on a 4095-token
    cache; Llama-2-13B H = 40 on 16383 tokens), bf16 — the oracle must reproduce output and stash; C2: also the prune of
    that stash (start 4 / important 1020 / recent 1024) under the tie rule."""
    g = _load(golden_dir, "g6_fullsize.npz")
    H, P, d, seed = (int(x) for x in g[f"{name}_meta"])
    dt = "bf16"
    q = orc.synth_normal(seed, 0, (1, H, 1, d), dt)
    k = orc.synth_normal(seed, 1, (1, H, 1, d), dt)
    v = orc.synth_normal(seed, 2, (1, H, 1, d), dt)
    past = (orc.synth_normal(seed, 3, (1, H, P, d), dt), orc.synth_normal(seed, 4, (1, H, P, d), dt))
    assert abs(_cksum(q, k, v, *past) - float(g[f"{name}_inck"])) < 1e-5
    o, stash, _ = orc.attention_core(q, k, v, past[0], past[1], np.full((1, 1), P), np.zeros((1, 1, 1, P + 1), np.float32), dt)
    want_stash = _bf16(g[f"{name}_stash"])
    got_stash = stash if name == "c2" else stash[:, ::5]
    np.testing.assert_allclose(got_stash, want_stash, **TOL[dt])
    assert np.mean(got_stash != want_stash) < 0.02
    np.testing.assert_allclose(o, _bf16(g[f"{name}_out"]), **TOL[dt])
    if name == "c2":
        kept = g["c2_kept"].astype(np.int64)                   # [32, 2048]: start | important (ascending) | recent
        N = P + 1
        assert np.array_equal(kept[:, :4], np.tile(np.arange(4), (H, 1))) and np.array_equal(kept[:, 1024:], np.tile(np.arange(N - 1024, N), (H, 1)))
        score = want_stash[0, :, 0]                            # importance = the stash itself at B = q = 1 (:51)
        idx = orc.topk_window(score, 4, N - 1024, 1020)
        tied = g["c2_tied_heads"]
        assert tied.sum() > 0, "the fixture is expected to contain threshold ties (3068 bf16 candidates per head)"
        for h in range(H):
            ref = kept[h, 4:1024]
            assert np.all(np.diff(ref) > 0)
            if not tied[h]:
                assert np.array_equal(idx[h], ref), h       # no tie at the threshold: bit exact
            assert tie_rule_equal(idx[h], ref, score[h], 4, 1020), h
