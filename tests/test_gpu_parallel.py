"""Head-parallel path through the HIP kernels on ONE GPU: every rank's shard is computed by the real kernels (one after
the other) and merged exactly as the all-gather lays the slices out — equal to the unsharded kernels; and the
library-owned RCCL communicator (spatten_comm_*) runs its all-gather, eagerly and captured in a HIP graph, at world
size 1 (a multi-GPU node is the driver's; the collective call, its layout and its capture are what can be checked here)."""
import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import TORCH_DT, attn_inputs, dev, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_decode_prune_equals_unsharded_on_the_hip_kernels(world):
    """BASELINE.json configs[2] partitioning: rank r owns heads [r*H/G, (r+1)*H/G) and their KV planes."""
    from spatten_amd import SpAttenKVCache, ops
    dt, B, H, d, N = "bf16", 1, 32, 128, 1500
    tdt = TORCH_DT[dt]
    q, k, v, past = attn_inputs(B, H, H, d, N - 1, 1, dt, seed=61)
    cos, sin = ops.rope_table(N + 8, d, tdt, "cuda")
    Kc = dev(np.concatenate([past[0], k], 2), dt)
    Vc = dev(np.concatenate([past[1], v], 2), dt)
    Kr = ops.rope_single(Kc, cos, sin)
    qd = dev(q[:, :, 0], dt)
    stash = torch.empty(B, H, N, dtype=tdt, device="cuda")
    head_abs = torch.zeros(B * H, dtype=torch.float32, device="cuda")
    full = ops.attn_decode(qd, None, Kr, Vc, N, cos, sin, N - 1, scores=stash, head_abs=head_abs)
    cache = SpAttenKVCache(4, 300, 500)
    new_full = cache.apply_token_pruning([(Kc, Vc)], 10, [stash[:, :, None, :]])
    idx_full = cache.keep_indices.clone()
    # --- the same work, one rank's shard at a time (what each process of a head-parallel run does) ---
    Hl = H // world
    staging = torch.empty(world, B, 1, Hl * d, dtype=tdt, device="cuda")        # all-gather receive layout: rank major
    scores_all = torch.empty(world, Hl, dtype=torch.float32, device="cuda")
    for r in range(world):
        lo, hi = r * Hl, (r + 1) * Hl
        Kl, Vl, Krl = Kc[:, lo:hi].contiguous(), Vc[:, lo:hi].contiguous(), Kr[:, lo:hi].contiguous()
        st_l = torch.empty(B, Hl, N, dtype=tdt, device="cuda")
        ha_l = torch.zeros(B * Hl, dtype=torch.float32, device="cuda")
        o_l = ops.attn_decode(qd[:, lo:hi].contiguous(), None, Krl, Vl, N, cos, sin, N - 1, scores=st_l, head_abs=ha_l)
        staging[r] = o_l.view(B, 1, Hl * d)
        scores_all[r] = ha_l.view(B, Hl).sum(0)
        assert torch.equal(st_l, stash[:, lo:hi])                                 # per-head work: bit identical
        c_l = SpAttenKVCache(4, 300, 500)
        new_l = c_l.apply_token_pruning([(Kl, Vl)], 10, [st_l[:, :, None, :]])
        assert torch.equal(c_l.keep_indices, idx_full[:, lo:hi])                  # token pruning needs no communication
        assert torch.equal(new_l[0][0], new_full[0][0][:, lo:hi]) and torch.equal(new_l[0][1], new_full[0][1][:, lo:hi])
    merged = staging.permute(1, 2, 0, 3).reshape(B, 1, H * d)                     # HeadParallel.gather_heads' merge
    # (a shard of H/G heads runs with more splits per head than the full launch: another fp32 merge order, <= 1 bf16 ulp)
    assert torch.allclose(merged[:, 0].float(), full.float(), atol=2e-3, rtol=1e-2)
    # head pruning: all-gathered H/G scores -> the same top-k on every rank
    np.testing.assert_allclose(host(scores_all.reshape(-1)), host(head_abs.view(B, H).sum(0)), rtol=1e-2)
    keep = ops.topk_select(scores_all.reshape(1, -1).contiguous(), 0, H, 24)[0]
    assert np.array_equal(keep.cpu().numpy(), orc.head_prune_select(host(scores_all.reshape(-1)), 24))


def test_native_rccl_allgather_eager_and_graph_captured():
    from spatten_amd import _lib
    from spatten_amd.parallel import HeadParallel
    hp = HeadParallel(32)
    assert hp.world == 1
    try:
        hp.init_native()
    except RuntimeError as e:                        # librccl missing on the box: the path must say so, not crash
        assert "unsupported" in str(e)
        pytest.skip("librccl not installed")
    send = torch.randn(32, 1, 512, device="cuda").to(torch.bfloat16)
    recv = torch.zeros_like(send)
    hp.allgather_native(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    # the collective is an ordinary stream operation of OUR communicator: it captures into a HIP graph and replays
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        recv.zero_()
        hp.allgather_native(send, recv)              # warm the capture path
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            send.mul_(2)
            hp.allgather_native(send, recv)
        want = send.clone()
        for _ in range(3):
            g.replay()
            want = want * 2
        side.synchronize()
    assert torch.equal(recv, want) and torch.equal(send, want)
    hp.close_native()
