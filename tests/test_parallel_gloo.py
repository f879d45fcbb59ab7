"""Head-parallel path with world_size 2 over gloo on CPU: sharding + all-gather reproduce the single-process
result (the per-rank compute is the oracle on the rank's heads — the HIP kernels are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import spatten_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spatten_amd.parallel import HeadParallel
        B, H, d, P, dt = 2, 8, 64, 100, "f32"
        q = orc.synth_normal(5, 0, (B, H, 1, d), dt)
        k = orc.synth_normal(5, 1, (B, H, 1, d), dt)
        v = orc.synth_normal(5, 2, (B, H, 1, d), dt)
        pk = orc.synth_normal(5, 3, (B, H, P, d), dt)
        pv = orc.synth_normal(5, 4, (B, H, P, d), dt)
        pos = np.full((B, 1), P)
        full_o, full_stash, _ = orc.attention_core(q, k, v, pk, pv, pos, None, dt)

        hp = HeadParallel(H)
        assert hp.world == world and hp.local_heads == H // world
        lo, hi = hp.head_range()
        sh = lambda x: x[:, lo:hi]
        o_loc, stash_loc, _ = orc.attention_core(sh(q), sh(k), sh(v), sh(pk), sh(pv), pos, None, dt)
        _, handle = hp.gather_heads(torch.from_numpy(o_loc), async_op=True)
        full = handle.wait()
        assert full.shape == (B, 1, H * d)
        # gather layout == the reference's head-major merge (values: BLAS blocks differ with the head count)
        np.testing.assert_allclose(full.numpy(), full_o, rtol=1e-5, atol=1e-6)
        full_sync, none = hp.gather_heads(torch.from_numpy(o_loc))
        assert none is None and torch.equal(full_sync, full)
        # token pruning is per head: local top-k == the rows of the global one, no communication
        imp = orc.importance(stash_loc, dt)
        idx_loc = orc.topk_window(imp, 4, P - 20, 30)
        idx_full = orc.topk_window(orc.importance(full_stash, dt), 4, P - 20, 30)
        np.testing.assert_array_equal(idx_loc, idx_full[lo:hi])
        # head pruning: all-gather of H/G head scores, identical top-k everywhere
        hs_full = orc.head_scores(full_o, H)
        hs = hp.gather_head_scores(torch.from_numpy(orc.head_scores(o_loc, hi - lo)))
        np.testing.assert_allclose(hs.numpy(), hs_full, rtol=1e-6)
        keep = orc.head_prune_select(hs.numpy(), 6)
        assert np.array_equal(keep, orc.head_prune_select(hs_full, 6))
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_head_parallel_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}
