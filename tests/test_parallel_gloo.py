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
        # cascade head pruning with STATIC ownership (BASELINE.json configs[2] / [4]; bench.py --config c3 / c5): the heads of
        # all ranks are ranked together, every rank keeps the local ids of its own survivors — uneven, here even none
        sc_full = np.array([[9, 8, 7, 6, 1, 2, 3, 5], [1, 1, 1, 1, 9, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
        keep = [5, 3, 2]
        want = orc.head_prune_cascade(list(sc_full), keep)
        got = hp.surviving_local_heads(torch.from_numpy(sc_full[:, lo:hi].copy()), keep)
        for l in range(3):
            mine = [int(h) - lo for h in want[l] if lo <= h < hi]
            assert got[l].dtype == torch.int32 and got[l].tolist() == mine, (rank, l, got[l].tolist(), mine)
        assert [len(g) for g in got] == ([4, 3, 2] if rank == 0 else [1, 0, 0])
        # global token scope: ONE kept set per layer, ranked by the importance summed over the heads of ALL ranks — the
        # all-reduce of the [layers, L] sums (SpAttenKVCache._global_scores) gives every rank the oracle's ranking
        from spatten_amd.kv_cache_token_pruning import SpAttenKVCache
        cache = SpAttenKVCache(start_size=4, recent_size=20, important_size=30, token_scope="global")
        cache.head_parallel = hp
        imp_full = orc.importance(full_stash, dt)
        rows = cache._global_scores([torch.from_numpy(imp), torch.from_numpy(imp * 0.5)])
        want = orc.global_token_scores(imp_full)
        assert rows[0].shape == (hi - lo, P + 1) and rows[0].stride(0) == 0
        np.testing.assert_array_equal(rows[0].numpy(), want[: hi - lo])
        np.testing.assert_array_equal(rows[1].numpy(), orc.global_token_scores(imp_full * 0.5)[: hi - lo])
        idx_g = orc.topk_window(rows[0].numpy(), 4, P - 20, 30)
        assert np.array_equal(idx_g, orc.topk_window(want, 4, P - 20, 30)[lo:hi])
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_head_parallel_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def _worker_c5(rank, world, port, ret):
    """BASELINE.json configs[4] partition on EIGHT ranks: 40 heads, 5 per rank, cascade head pruning to 30 with static ownership —
    the per-rank head lists `bench.py --config c5 --gpus 8` launches — plus the gather layout at world 8."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spatten_amd.parallel import HeadParallel
        B, H, d, P, dt = 1, 40, 16, 37, "f32"
        q = orc.synth_normal(15, 0, (B, H, 1, d), dt)
        k = orc.synth_normal(15, 1, (B, H, 1, d), dt)
        v = orc.synth_normal(15, 2, (B, H, 1, d), dt)
        pk = orc.synth_normal(15, 3, (B, H, P, d), dt)
        pv = orc.synth_normal(15, 4, (B, H, P, d), dt)
        pos = np.full((B, 1), P)
        full_o, _, _ = orc.attention_core(q, k, v, pk, pv, pos, None, dt)
        hp = HeadParallel(H)
        assert hp.world == world == 8 and hp.local_heads == 5
        lo, hi = hp.head_range()
        assert (lo, hi) == (5 * rank, 5 * rank + 5)
        sh = lambda x: x[:, lo:hi]
        o_loc, _, _ = orc.attention_core(sh(q), sh(k), sh(v), sh(pk), sh(pv), pos, None, dt)
        full, _ = hp.gather_heads(torch.from_numpy(o_loc))
        np.testing.assert_allclose(full.numpy(), full_o, rtol=1e-5, atol=1e-6)          # rank-major = head-major
        # head scores of three layers, pruned 40 -> 30 -> 30 -> 24 (cumulative: a pruned head stays pruned)
        rng = np.random.default_rng(7)
        sc_full = rng.random((3, H)).astype(np.float32)
        keep = [30, 30, 24]
        want = orc.head_prune_cascade(list(sc_full), keep)
        got = hp.surviving_local_heads(torch.from_numpy(sc_full[:, lo:hi].copy()), keep)
        counts = []
        for l in range(3):
            mine = [int(h) - lo for h in want[l] if lo <= h < hi]
            assert got[l].tolist() == mine, (rank, l, got[l].tolist(), mine)
            counts.append(len(mine))
        tot = torch.tensor(counts)
        dist.all_reduce(tot)
        assert tot.tolist() == keep                                                       # every survivor has exactly one owner
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_head_parallel_world8_c5_partition_gloo():
    world, port = 8, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_c5, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: 1 for r in range(8)}


def _plugin_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace

        from torch import nn

        from spatten_amd.parallel import HeadParallel
        from spatten_amd.pos_shift.modify_llama import shard_attention_projections
        torch.manual_seed(3)                      # the same random Llama-attention stub on every rank
        B, H, Hkv, d, P, dt = 2, 8, 4, 16, 40, "f32"
        HID = H * d

        class LlamaAttention(nn.Module):
            def __init__(self):
                super().__init__()
                self.config = SimpleNamespace(pretraining_tp=1)
                self.num_heads, self.num_key_value_heads, self.head_dim, self.hidden_size = H, Hkv, d, HID
                self.q_proj = nn.Linear(HID, H * d, bias=True)
                self.k_proj = nn.Linear(HID, Hkv * d, bias=True)
                self.v_proj = nn.Linear(HID, Hkv * d, bias=True)
                self.o_proj = nn.Linear(HID, HID, bias=False)
        m = LlamaAttention()
        hp = HeadParallel(H, Hkv)
        x = torch.from_numpy(orc.synth_normal(9, 0, (B, 1, HID), dt))
        pk = orc.synth_normal(9, 3, (B, Hkv, P, d), dt)
        pv = orc.synth_normal(9, 4, (B, Hkv, P, d), dt)
        pos = np.full((B, 1), P)
        heads = lambda t, n: t.detach().view(B, 1, n, d).transpose(1, 2).numpy()
        with torch.no_grad():
            # unsharded: the reference op sequence on all heads (oracle), then o_proj
            o_full, _, _ = orc.attention_core(heads(m.q_proj(x), H), heads(m.k_proj(x), Hkv), heads(m.v_proj(x), Hkv), pk, pv, pos, None, dt)
            y_full = m.o_proj(torch.from_numpy(o_full))
            for fuse in (False, True):
                shard_attention_projections(m, hp, fuse=fuse)
                hpo, parts, stacked = m.__dict__["_spatten_hp"]
                assert hpo is hp and (stacked is not None) == fuse
                Hl, Hkvl = hp.local_heads, hp.local_kv_heads
                assert parts[0][0].shape == (Hl * d, HID) and parts[1][0].shape == (Hkvl * d, HID) and parts[0][1].shape == (Hl * d,)
                if fuse:
                    qkv = torch.nn.functional.linear(x, stacked[0], stacked[1])
                    ql, kl, vl = qkv[..., :stacked[2]], qkv[..., stacked[2]:stacked[2] + stacked[3]], qkv[..., stacked[2] + stacked[3]:]
                else:
                    ql, kl, vl = (torch.nn.functional.linear(x, w, b) for w, b in parts)
                lo, hi = hp.kv_head_range()
                # the rank's column-sharded projections give exactly its heads of the full projections
                hl, hh = hp.head_range()
                np.testing.assert_allclose(heads(ql, Hl), heads(m.q_proj(x), H)[:, hl:hh], rtol=1e-5, atol=1e-6)
                o_loc, _, _ = orc.attention_core(heads(ql, Hl), heads(kl, Hkvl), heads(vl, Hkvl), pk[:, lo:hi], pv[:, lo:hi], pos, None, dt)
                full, _ = hp.gather_heads(torch.from_numpy(o_loc))          # the exchange in front of o_proj (:146-163)
                y = m.o_proj(full)
                np.testing.assert_allclose(y.numpy(), y_full.numpy(), rtol=1e-4, atol=1e-5)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_plugin_head_parallel_slicing_and_gather_layout_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_plugin_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_comm_entry_points_reject_bad_arguments_and_a_missing_rccl():
    """spatten_comm_* failure modes that need no GPU: argument checks, and SPATTEN_ERR_UNSUPPORTED when RCCL cannot be
    loaded (SPATTEN_RCCL_LIB points the loader at a path; a fresh process, because the loader runs once)."""
    import subprocess
    import sys
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
from spatten_amd import _lib
lib = _lib.load()
ident = ctypes.create_string_buffer(128)
comm = ctypes.c_void_p()
assert lib.spatten_comm_unique_id(None) == -1
assert lib.spatten_comm_init(ctypes.byref(comm), 2, 2, ident) == -1          # rank outside [0, nranks)
assert lib.spatten_comm_init(ctypes.byref(comm), 0, 0, ident) == -1
assert lib.spatten_comm_init(None, 0, 1, ident) == -1
assert lib.spatten_comm_init(ctypes.byref(comm), 0, 1, None) == -1
assert lib.spatten_allgather(None, ident, ident, 8, None) == -1
n = ctypes.c_int()
assert lib.spatten_comm_info(None, ctypes.byref(n), ctypes.byref(n)) == -1
assert lib.spatten_comm_destroy(None) == 0
assert lib.spatten_comm_unique_id(ident) == -2, "RCCL must be reported missing"     # SPATTEN_ERR_UNSUPPORTED
assert lib.spatten_comm_init(ctypes.byref(comm), 0, 1, ident) == -2
from spatten_amd.parallel import HeadParallel
try:
    HeadParallel(8).init_native()
    raise SystemExit("init_native must raise")
except RuntimeError as e:
    assert "unsupported" in str(e), str(e)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPATTEN_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def _handles_worker(rank, world, port, q):
    import torch.distributed as dist
    from spatten_amd.parallel import HeadParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hp = HeadParallel(8)
        mine = bytes([rank + 1]) * 64                       # stands in for this rank's hipIpc window handle
        got = hp.gather_handles(mine)
        q.put((rank, [h[0] for h in got], all(len(h) == 64 for h in got)))
    finally:
        dist.destroy_process_group()


def test_peer_store_handle_exchange_is_rank_major_world2_gloo():
    """The out-of-band step of the peer-store all-gather (include/spatten.h: spatten_peer_connect takes the handles rank-major):
    every rank's 64-byte handle reaches every rank in rank order through torch.distributed (CPU, two processes)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_handles_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, [1, 2], True), (1, [1, 2], True)]
