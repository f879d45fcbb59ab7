"""Seeded random sweeps of the three hot-path pieces against the oracle: shapes, GQA ratios, batch, lengths that do not
align with any tile or split boundary, dtypes.  Complements the hand-picked cases of the other GPU tests."""
import os

import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.test_gpu_decode import run_decode
from tests.test_gpu_prefill import run_prefill
from tests.util import OUT_TOL, attn_inputs, check_stash, dev, host

pytestmark = pytest.mark.gpu

# SPATTEN_SWEEP_SEEDS=n widens every sweep to n seeds (soak runs; the default keeps the suite short)
SEEDS = int(os.environ.get("SPATTEN_SWEEP_SEEDS", "0"))


@pytest.mark.parametrize("seed", range(max(10, SEEDS)))
def test_decode_random_configs(seed):
    rng = np.random.default_rng(1000 + seed)
    dt = str(rng.choice(["f32", "bf16", "f16"]))
    d = int(rng.choice([64, 128]))
    Hkv = int(rng.choice([1, 2, 4]))
    H = Hkv * int(rng.choice([1, 2, 4]))
    B = int(rng.choice([1, 2, 3]))
    P = int(rng.integers(0, 2500))
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, 1, dt, seed=7000 + seed)
    pos = np.full((B, 1), P)
    o, stash, (kc, vc) = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos, None, dt)
    splits = int(rng.choice([0, 0, 1, 2, 3, 7]))
    out, st, kc_g, vc_g, _ = run_decode(q, k, v, past, dt, n_splits=splits)
    np.testing.assert_allclose(out, o, err_msg=f"{dt} d{d} H{H}/{Hkv} B{B} P{P} S{splits}", **OUT_TOL[dt])
    check_stash(st, stash, dt)
    assert np.array_equal(kc_g, kc) and np.array_equal(vc_g, vc)


@pytest.mark.parametrize("seed", range(max(8, SEEDS)))
def test_prefill_random_configs(seed):
    rng = np.random.default_rng(2000 + seed)
    dt = str(rng.choice(["bf16", "f16"]))
    d = int(rng.choice([64, 128]))
    Hkv = int(rng.choice([1, 2]))
    H = Hkv * int(rng.choice([1, 2, 4]))
    B = int(rng.choice([1, 2]))
    P = int(rng.integers(0, 700 if seed < 8 else 3000))      # later seeds: long pasts (key-split flash kernel)
    ql = int(rng.integers(9, 600))
    q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=8000 + seed)
    N = P + ql
    pos = np.tile(np.arange(P, N)[None], (B, 1))
    o, stash, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos,
                                     orc.causal_mask(B, ql, N, dt), dt)
    msg = f"{dt} d{d} H{H}/{Hkv} B{B} P{P} q{ql}"
    out, _, _ = run_prefill(q, k, v, past, dt, causal=True, stash=False)
    np.testing.assert_allclose(out, o, err_msg=msg, **OUT_TOL[dt])
    out, st, ci = run_prefill(q, k, v, past, dt, causal=True, stash=True, colimp=True)
    np.testing.assert_allclose(out, o, err_msg=msg, **OUT_TOL[dt])
    check_stash(st, stash, dt, msg)
    np.testing.assert_allclose(ci, st.sum(axis=2, dtype=np.float32), rtol=1e-4, atol=2e-3, err_msg=msg)


@pytest.mark.parametrize("seed", range(max(8, SEEDS)))
def test_prune_random_configs(seed):
    from spatten_amd import SpAttenKVCache
    rng = np.random.default_rng(3000 + seed)
    dt = str(rng.choice(["f32", "bf16", "f16"]))
    d = int(rng.choice([16, 64, 128]))
    H = int(rng.integers(1, 9))
    B = int(rng.choice([1, 2]))
    L = int(rng.integers(200, 3000))
    start = int(rng.integers(0, 8))
    recent = int(rng.integers(8, L // 3))
    important = int(rng.integers(1, L // 3))
    coming = int(rng.integers(0, recent + 1))
    nl = int(rng.choice([1, 3]))
    K = [orc.synth_normal(9000 + seed, 10 + l, (B, H, L, d), dt) for l in range(nl)]
    V = [orc.synth_normal(9000 + seed, 20 + l, (B, H, L, d), dt) for l in range(nl)]
    S = [orc.synth_normal(9000 + seed, 30 + l, (B, H, 1, L), dt) for l in range(nl)]
    cache = SpAttenKVCache(start, recent, important)
    past = [(dev(K[l], dt), dev(V[l], dt)) for l in range(nl)]
    out = cache.apply_token_pruning(past, coming, [dev(S[l], dt) for l in range(nl)])
    torch.cuda.synchronize()
    want, idx = orc.apply_token_pruning([(K[l], V[l]) for l in range(nl)], coming, S, start, recent, important, dt)
    assert idx is not None                                  # the ranges above always leave something to prune
    for l in range(nl):
        assert np.array_equal(host(out[l][0]), want[l][0]) and np.array_equal(host(out[l][1]), want[l][1]), (dt, d, H, B, L, start, recent, important, coming, l)
