"""Pin the plain-C oracle (oracle/oracle.c) against the reference goldens and the numpy oracle.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, attn_inputs, check_stash, golden


def test_c_decode_matches_reference_goldens():
    g = golden("g3_attention.npz")
    n = 0
    for m in g["meta"]:
        name, B, H, Hkv, d, P, ql, mask_kind, dt, seed = m.split("|")
        B, H, Hkv, d, P, ql, seed = map(int, (B, H, Hkv, d, P, ql, seed))
        if ql != 1:
            continue
        q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed)
        kc = k if past is None else np.concatenate([past[0], k], 2)
        vc = v if past is None else np.concatenate([past[1], v], 2)
        cos, sin = orc.rope_table(P + 1, d, dt)
        mask = np.zeros((B, P + 1), np.float32) if mask_kind == "zeros" else None
        out, stash = co.attn_decode(q[:, :, 0], kc, vc, cos[:, : d // 2], sin[:, : d // 2], P, dt, mask=mask)
        # the C port rounds P to the model dtype like the reference does -> tight tolerance
        np.testing.assert_allclose(out, g[f"{name}_out"], err_msg=name, **OUT_TOL[dt])
        check_stash(stash, g[f"{name}_stash"], dt, name)
        n += 1
    assert n >= 10


def test_c_prune_matches_reference_goldens():
    g = golden("g1_prune.npz")
    for m in g["meta"]:
        name, H, L, d, start, recent, important, c, qs, dt, seed, bump = m.split("|")
        H, L, d, start, recent, important, c, qs, seed, bump = map(int, (H, L, d, start, recent, important, c, qs, seed, bump))
        stash = orc.synth_normal(seed + 1000 * bump, 5, (1, H, qs, L), dt)
        K = orc.synth_normal(seed, 6, (1, H, L, d), dt)
        V = orc.synth_normal(seed, 7, (1, H, L, d), dt)
        score = orc.importance(stash, dt)
        hi = min(L - recent + c, L)
        idx = co.topk_window(score, start, hi, important, dt)
        assert np.array_equal(idx, orc.topk_window(score, start, hi, important)), name
        Kn = co.from_raw(co.kv_compact_raw(dt, co.to_raw(K, dt), idx, start, hi), dt)
        Vn = co.from_raw(co.kv_compact_raw(dt, co.to_raw(V, dt), idx, start, hi), dt)
        assert np.array_equal(Kn, g[f"{name}_K"]) and np.array_equal(Vn, g[f"{name}_V"]), name


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_c_conversions_round_trip_and_rne(dt):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-8, 6, 20000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 5.96e-8, 3e-8, np.inf, -np.inf], np.float32)])
    want = orc.round_dt(x, dt)
    got = co.from_raw(co.to_raw(want, dt), dt)
    assert np.array_equal(got, want)
    # the C rounding itself (through a 1-key "attention": stash of q.k with d=64 is exercised above); here
    # check top-k's tie rule + NaN order in C
    s = np.array([[1, 3, 3, 3, 2, 3, 0, 3]], np.float32)
    assert co.topk_window(s, 0, 8, 2).tolist() == [[1, 2]]
    s = np.array([[0.0, -0.0, np.nan, 1.0, -np.inf, np.inf]], np.float32)
    assert co.topk_window(s, 0, 6, 3).tolist() == [[2, 3, 5]]
    with pytest.raises(ValueError):
        co.topk_window(np.zeros((1, 10), np.float32), 4, 8, 5)


def test_c_decode_vs_numpy_oracle_mask_and_gqa():
    dt, B, H, Hkv, d, P = "bf16", 2, 8, 2, 128, 200
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, 1, dt, seed=9)
    N = P + 1
    rng = np.random.default_rng(1)
    mask = np.where(rng.random((B, 1, 1, N)) < 0.3, np.float32(orc.finfo_min(dt)), np.float32(0)).astype(np.float32)
    mask[..., -1] = 0
    o, stash, (kc, vc) = orc.attention_core(q, k, v, past[0], past[1], np.full((B, 1), P), mask, dt)
    cos, sin = orc.rope_table(N, d, dt)
    out, st = co.attn_decode(q[:, :, 0], kc, vc, cos[:, : d // 2], sin[:, : d // 2], P, dt, mask=mask[:, 0, 0])
    np.testing.assert_allclose(out, o, **OUT_TOL[dt])
    check_stash(st, stash, dt)
