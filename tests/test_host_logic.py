"""Host-side logic of the drop-in surface (no GPU): window arithmetic, passthrough rules, errors, slabs, sharding."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from spatten_amd import SpAttenKVCache, kv_slab
from spatten_amd.parallel import HeadParallel


def test_constructor_surface_matches_reference(capsys):
    c = SpAttenKVCache()                                           # defaults kv_cache_token_pruning.py:26-30
    assert (c.start_size, c.recent_size, c.important_size, c.cache_size, c.k_seq_dim, c.v_seq_dim) == (4, 128, 128, 260, 2, 2)
    assert "SpAttenKVCache: keep start: 4, keep recent: 128, keep important: 128" in capsys.readouterr().out
    x = torch.arange(2 * 3 * 10 * 4).reshape(2, 3, 10, 4)
    assert torch.equal(c.k_slice(x, 2, 5), x[:, :, 2:5])


def test_passthrough_and_none_need_no_gpu():
    c = SpAttenKVCache(4, 32, 28)
    assert c.apply_token_pruning(None, 3, []) is None
    past = [(torch.zeros(1, 2, 40, 8), torch.zeros(1, 2, 40, 8))]
    assert c.apply_token_pruning(past, 24, [torch.zeros(1, 2, 1, 40)]) is past      # 40 + 24 <= 64
    with pytest.raises(RuntimeError, match="ROCm device tensors"):                     # pruning branch: HIP only
        c.apply_token_pruning(past, 25, [torch.zeros(1, 2, 1, 40)])


@pytest.mark.parametrize("L,c", [(300, 20), (300, 100), (4096, 0), (4096, 64), (128, 0)])
def test_window_matches_oracle_shapes(L, c):
    s, r, m = 4, 64, 50
    cache = SpAttenKVCache(s, r, m)
    lo, hi, tail_lo, new_len = cache.window(L, c)
    assert (lo, hi) == (s, min(L - r + c, L))
    past = [(np.zeros((1, 2, L, 4), np.float32), np.zeros((1, 2, L, 4), np.float32))]
    stash = [orc.synth_normal(1, 0, (1, 2, 1, L), "f32")]
    new, _ = orc.apply_token_pruning(past, c, stash, s, r, m, "f32")
    assert new[0][0].shape[2] == new_len == s + m + max(0, r - c)


def test_errors_replace_reference_crashes():
    past = [(torch.zeros(1, 2, 40, 8), torch.zeros(1, 2, 40, 8))]
    with pytest.raises(ValueError, match="important_size"):          # reference: TypeError at :63
        SpAttenKVCache(4, 8, 0).apply_token_pruning(past, 30, [torch.zeros(1, 2, 1, 40)])
    with pytest.raises(ValueError, match="fewer than important_size"):   # reference: torch.topk RuntimeError
        SpAttenKVCache(4, 8, 50).apply_token_pruning(past, 30, [torch.zeros(1, 2, 1, 40)])
    with pytest.raises(NotImplementedError):
        SpAttenKVCache(4, 8, 8, k_seq_dim=1, v_seq_dim=1).apply_token_pruning(
            [(torch.zeros(1, 40, 8), torch.zeros(1, 40, 8))], 30, [torch.zeros(1, 2, 1, 40)])


def test_slab_capacity_rounding():
    assert kv_slab.round_capacity(1) == 128 and kv_slab.round_capacity(2048 + 64) == 2176 and kv_slab.round_capacity(128) == 128


def test_head_parallel_ranges_single_process():
    hp = HeadParallel(32, 8)
    assert hp.world == 1 and hp.head_range() == (0, 32) and hp.kv_head_range() == (0, 8)
    x = torch.arange(2 * 32 * 3).reshape(2, 32, 3)
    assert torch.equal(hp.shard_heads(x), x)
    full, work = hp.gather_heads(torch.zeros(2, 1, 64))
    assert work is None and full.shape == (2, 1, 64)
    none, handle = hp.gather_heads(torch.zeros(2, 1, 64), async_op=True)
    assert none is None and handle.wait().shape == (2, 1, 64)


def test_bf16_logit_scale_by_reciprocal_is_exact():
    """The flash kernel scales a bf16 logit by 1/sqrt(128) with ONE multiply (csrc/common.h logit_scale): claimed equal
    to the reference's fp32 division followed by the bf16 rounding for every bf16 input."""
    import math
    import numpy as np
    from oracle import spatten_oracle as orc
    c = np.float32(math.sqrt(128))
    rc = np.float32(1.0) / c
    m = np.arange(128, 256).astype(np.float32)
    for e in range(-100, 100):
        x = orc.round_dt(m * np.float32(2.0) ** np.float32(e), "bf16")
        for sgn in (1.0, -1.0):
            xs = (x * np.float32(sgn)).astype(np.float32)
            assert np.array_equal(orc.round_dt((xs / c).astype(np.float32), "bf16"), orc.round_dt((xs * rc).astype(np.float32), "bf16")), e


def test_global_token_scope_rows_and_validation():
    """token_scope="global" (README.md:21; workloads/small.csv:1): the select is handed H stride-0 rows of the head-summed
    importance (fp64 sum, one rounding to fp32) — pure host logic, equal to the oracle's restatement."""
    import numpy as np
    from oracle import spatten_oracle as orc
    with pytest.raises(ValueError):
        SpAttenKVCache(token_scope="layer")
    cache = SpAttenKVCache(start_size=4, recent_size=8, important_size=8, token_scope="global")
    assert cache.token_scope == "global" and cache.head_parallel is None
    sc = [orc.synth_normal(3 + l, 1, (6, 50), "f32") for l in range(2)]
    rows = cache._global_scores([torch.from_numpy(s) for s in sc])
    for l in range(2):
        assert rows[l].shape == (6, 50) and rows[l].stride() == (0, 1) and rows[l].dtype == torch.float32
        np.testing.assert_array_equal(rows[l].numpy(), orc.global_token_scores(sc[l]))
    # an explicit two-rank partition sums through reduce_fn
    other = torch.from_numpy(np.stack([s[3:].astype(np.float64).sum(0) for s in sc]))
    cache.head_parallel = HeadParallel(6, rank=0, world=2, gather_fn=lambda t, r: t, reduce_fn=lambda t, r: t + other)
    rows = cache._global_scores([torch.from_numpy(s[:3]) for s in sc])
    np.testing.assert_array_equal(rows[1].numpy(), orc.global_token_scores(sc[1])[:3])
