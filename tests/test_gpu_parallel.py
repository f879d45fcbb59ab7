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
    assert hp.native_info() == (1, 0)                # what the RCCL communicator itself reports
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


# ------------------------------------------------------------------------------------------------------------------
# head-parallel mode of the PLUGIN: enable_spatten_llm(..., head_parallel=hp)
# ------------------------------------------------------------------------------------------------------------------
def _stack(layers, H, Hkv, d, dt, bias=False):
    from types import SimpleNamespace

    from torch import nn

    class LlamaAttention(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(pretraining_tp=1)
            self.num_heads, self.num_key_value_heads, self.num_key_value_groups = H, Hkv, H // Hkv
            self.head_dim, self.hidden_size = d, H * d
            self.q_proj = nn.Linear(H * d, H * d, bias=bias, dtype=dt, device="cuda")
            self.k_proj = nn.Linear(H * d, Hkv * d, bias=bias, dtype=dt, device="cuda")
            self.v_proj = nn.Linear(H * d, Hkv * d, bias=bias, dtype=dt, device="cuda")
            self.o_proj = nn.Linear(H * d, H * d, bias=bias, dtype=dt, device="cuda")

    class Stack(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(model_type="llama")
            self.layers = nn.ModuleList([LlamaAttention() for _ in range(layers)])
    return Stack()


def _hf_args(x, P):
    B, q, _ = x.shape
    N = P + q
    pos = torch.arange(P, N, device=x.device)[None]
    mask = torch.zeros(B, 1, q, N, dtype=x.dtype, device=x.device)
    if q > 1:
        mask.masked_fill_(torch.ones(q, N, dtype=torch.bool, device=x.device).triu(P + 1), torch.finfo(x.dtype).min)
    return mask, pos


@pytest.mark.parametrize("dt,world,Hkv,kw", [
    (torch.float32, 2, 8, {}), (torch.float32, 4, 4, dict(fuse_qkv=True)), (torch.float32, 2, 8, dict(native_gemv=True, fuse_qkv=True)),
    (torch.float32, 2, 8, dict(head_keep=[6, 5])), (torch.bfloat16, 2, 8, dict(native_gemv=True)),
    (torch.float32, 2, 8, dict(importance_mode="cascade")), (torch.float32, 2, 8, dict(pq_threshold=0.05)),
    (torch.float32, 4, 4, dict(importance_mode="cascade", head_keep=[6, 5]))])
def test_plugin_head_parallel_shards_run_in_sequence_equal_the_unsharded_plugin(dt, world, Hkv, kw):
    """Every rank's copy of the patched stack (its column-sharded q/k/v projections, its H/G heads of KV cache) is driven
    layer by layer in ONE process; the all-gather is a loopback that lays the slices out rank-major.  Prefill, decode
    steps, a prune event and a second turn must reproduce the unsharded plugin: hidden states of every forward, the local
    caches = the head slices of the full caches, the same kept positions.  (A sharded projection is another GEMM shape,
    so its rows can differ from the full GEMM's in the last bits: fp32 runs the whole protocol with exact kept sets and
    caches to 1e-5; bf16 checks the first turn within rounding and the prune by agreement.)"""
    import contextlib
    import io

    from spatten_amd import enable_spatten_llm
    from spatten_amd.parallel import HeadParallel
    torch.manual_seed(7)
    L, H, d = 2, 8, 64
    exact = dt == torch.float32
    close = (lambda a, b: torch.allclose(a, b, atol=1e-5, rtol=1e-5)) if exact else (lambda a, b: torch.allclose(a.float(), b.float(), atol=3e-2, rtol=3e-2))
    HID = H * d
    full = _stack(L, H, Hkv, d, dt, bias="fuse_qkv" not in kw)
    for p in full.parameters():
        p.data.mul_(0.5)
    ranks = []
    shared = {}

    score_calls = {}

    def loopback(local, rank):            # the all-gather: rank r's slice lands at [r*n, (r+1)*n) of the last axis
        n = local.shape[-1]
        if n == H // world and local.dtype == torch.float32 and local.numel() == n:
            # head pruning's exchange of H/G cumulative scores: EVERY rank needs the full vector at its own prune, so the
            # loopback asks the other ranks' state for their share (a pure function of each rank's accumulators)
            layer = score_calls.get(rank, 0)
            score_calls[rank] = layer + 1
            return torch.cat([c.ext.head_scores(layer) for _, c, _ in ranks]).reshape(1, 1, -1)
        buf = shared.setdefault((tuple(local.shape), local.dtype), torch.zeros(*local.shape[:-1], n * world, dtype=local.dtype,
                                                                               device=local.device))
        buf[..., rank * n:(rank + 1) * n] = local
        return buf.clone()
    with contextlib.redirect_stdout(io.StringIO()):
        cache_full = enable_spatten_llm(full, 4, 40, 48, **kw)
        for r in range(world):
            m = _stack(L, H, Hkv, d, dt, bias="fuse_qkv" not in kw)
            m.load_state_dict(full.state_dict())
            hp = HeadParallel(H, Hkv, rank=r, world=world, gather_fn=loopback)
            ranks.append((m, enable_spatten_llm(m, 4, 40, 48, head_parallel=hp, **kw), hp))
    g = torch.Generator(device="cuda").manual_seed(8)
    tol = dict(atol=1e-4, rtol=1e-4) if exact else dict(atol=3e-2, rtol=3e-2)

    def forward_all(x, past_full, past_ranks):
        """one forward of the stack: unsharded, and every rank layer by layer (the LAST rank to run a layer sees the
        complete gathered tensor: its o_proj output is that layer's output on every rank of a real run)"""
        P = 0 if past_full is None else past_full[0][0].shape[2]
        mask, pos = _hf_args(x, P)
        xf, new_full = x, []
        xs, new_ranks = x, [[] for _ in range(world)]
        for i in range(L):
            a, _, kv = full.layers[i](xf, attention_mask=mask, position_ids=pos, past_key_value=None if past_full is None else past_full[i], use_cache=True)
            xf = xf + a
            new_full.append(kv)
            for r, (m, _, hp) in enumerate(ranks):
                ar, _, kvr = m.layers[i](xs, attention_mask=mask, position_ids=pos, past_key_value=None if past_ranks is None else past_ranks[r][i], use_cache=True)
                new_ranks[r].append(kvr)
            xs = xs + ar
            np.testing.assert_allclose(host(xs), host(xf), err_msg=f"layer {i}", **tol)
            xs = xf                       # keep the two runs on identical inputs: differences must not accumulate into the caches
        return new_full, new_ranks
    x0 = torch.randn(1, 120, HID, device="cuda", generator=g).to(dt)
    pf, pr = forward_all(x0, None, None)
    for turn in range(2):
        for t in range(6):
            pf, pr = forward_all(torch.randn(1, 1, HID, device="cuda", generator=g).to(dt), pf, pr)
        for r, (m, _, hp) in enumerate(ranks):
            lo, hi = hp.kv_head_range()
            for i in range(L):
                assert pr[r][i][0].shape[1] == Hkv // world
                if "head_keep" in kw and turn == 1:
                    continue          # pruned heads are not launched: their cache rows are never written (on either side)
                assert close(pr[r][i][0], pf[i][0][:, lo:hi]) and close(pr[r][i][1], pf[i][1][:, lo:hi])
                hl, hh = hp.head_range()
                if "head_keep" not in kw:
                    assert close(m.layers[i].attn_scores, full.layers[i].attn_scores[:, hl:hh])
        if turn == 0:
            coming = 10
            if "head_keep" in kw and Hkv != H:
                break
            new_f = cache_full.apply_token_pruning(pf, coming, [m.attn_scores for m in full.layers])
            assert new_f is not pf
            new_r = []
            score_calls.clear()
            for r, (m, cache_r, hp) in enumerate(ranks):
                nr = cache_r.apply_token_pruning(pr[r], coming, [mm.attn_scores for mm in m.layers])
                lo, hi = hp.kv_head_range()
                hl, hh = hp.head_range()
                for i in range(L):
                    if exact:         # token pruning is per head: the rank keeps exactly the rows the full model keeps for its heads
                        assert torch.equal(cache_r.keep_indices[i], cache_full.keep_indices[i][lo:hi])   # rows = KV heads
                        assert close(nr[i][0], new_f[i][0][:, lo:hi]) and close(nr[i][1], new_f[i][1][:, lo:hi])
                    else:
                        a_, b_ = cache_r.keep_indices[i].cpu().numpy(), cache_full.keep_indices[i][lo:hi].cpu().numpy()
                        for ra, rb in zip(a_, b_):      # rounding-level score differences can only swap threshold tokens
                            assert len(np.intersect1d(ra, rb)) >= 0.9 * len(ra)
                if "head_keep" in kw:       # the same global kept set on every rank, its local share launched
                    for i in range(L):
                        kg_f = cache_full.ext.layers[i].kept_global
                        kg_r = cache_r.ext.layers[i].kept_global
                        assert (kg_f is None and kg_r is None) or torch.equal(kg_f, kg_r)
                        hl, hh = hp.head_range()
                        if kg_f is not None:
                            want = [int(h) - hl for h in kg_f.tolist() if hl <= int(h) < hh]
                            assert cache_r.ext.layers[i].head_ids.tolist() == want
                new_r.append(nr)
            if not exact:
                break
            pf, pr = new_f, new_r
            pf, pr = forward_all(torch.randn(1, coming, HID, device="cuda", generator=g).to(dt), pf, pr)


def test_comm_init_refuses_two_ranks_on_one_device():
    """spatten_comm_init failure mode: both ranks of a 2-rank communicator on the SAME device — RCCL refuses ("duplicate
    GPU"), the entry point returns SPATTEN_ERR_LAUNCH on both instead of building a broken communicator.  Run in a child
    process under a timeout (two threads = the two ranks)."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes, sys, threading
sys.path.insert(0, %r)
import torch
torch.cuda.init()
from spatten_amd import _lib
lib = _lib.load()
ident = ctypes.create_string_buffer(128)
rc = lib.spatten_comm_unique_id(ident)
if rc == -2:
    print("norccl"); raise SystemExit(0)
assert rc == 0
res = [None, None]
def run(r):
    torch.cuda.set_device(0)
    comm = ctypes.c_void_p()
    res[r] = lib.spatten_comm_init(ctypes.byref(comm), r, 2, ident)
ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
print("rc", res[0], res[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.fail("spatten_comm_init with two ranks on one device did not return")
    if "norccl" in out.stdout:
        pytest.skip("librccl not installed")
    last = [ln for ln in out.stdout.splitlines() if ln.startswith("rc ")]
    assert last, (out.stdout[-500:], out.stderr[-1500:])
    a, b = (int(x) for x in last[-1].split()[1:])
    assert a == -4 and b == -4, last[-1]
