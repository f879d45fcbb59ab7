"""The chained decode launch (ABI 5, round 6; `spatten_attn_decode_chain`): the attention step of ALL layers of a token in
ONE launch must leave exactly what the per-layer launches (`spatten_attn_decode_args`, modify_llama.py:86-147 once per
module) leave — attention outputs, stash, appended cache rows, bit for bit — in its static and its device-length form, under
a captured graph, with per-layer head lists, at the C2 and the C5 geometry.  Needs an MI355X."""
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import OUT_TOL, TORCH_DT, attn_inputs, dev, host

pytestmark = pytest.mark.gpu


def _layers(L, B, H, d, P, cap, dt, seed, cos, sin):
    """Two identical sets of per-layer planes (past length P, finite garbage beyond), queries and new rows."""
    from spatten_amd import ops
    tdt = TORCH_DT[dt]
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(tdt)
    sets = []
    base = []
    for l in range(L):
        kc = torch.zeros(B, H, cap, d, dtype=tdt, device="cuda")
        vc, krc = torch.zeros_like(kc), torch.zeros_like(kc)
        kc[:, :, :P], vc[:, :, :P] = rnd(B, H, P, d), rnd(B, H, P, d)
        ops.build_shadow(kc, krc, 0, P, cos, sin)
        base.append((kc, krc, vc))
    for _ in range(2):
        sets.append([tuple(t.clone() for t in base[l]) for l in range(L)])
    return sets, rnd


def _run_pair(L, B, H, d, P, cap, dt, steps, hids=None, dyn=False, graph=False, seed=3, n_splits=0):
    """`steps` consecutive tokens through the per-layer launches (set A) and through the chain (set B); everything the
    step leaves must be equal."""
    from spatten_amd import ops
    tdt = TORCH_DT[dt]
    cos, sin = ops.rope_table(cap + 8, d, tdt, "cuda")
    (A, Bs), rnd = _layers(L, B, H, d, P, cap, dt, seed, cos, sin)
    q = [rnd(B, H, d) for _ in range(L)]
    kn = [rnd(B, H, d) for _ in range(L)]
    vn = [rnd(B, H, d) for _ in range(L)]
    out_a = [torch.zeros(B, H * d, dtype=tdt, device="cuda") for _ in range(L)]
    out_b = [torch.zeros_like(x) for x in out_a]
    st_a = [torch.zeros(B, H, cap, dtype=tdt, device="cuda") for _ in range(L)]
    st_b = [torch.zeros_like(x) for x in st_a]
    ws = ops.DecodeWorkspace(B, H, d, "cuda")
    chain = ops.DecodeChain(q, [x[0] for x in Bs], [x[1] for x in Bs], [x[2] for x in Bs], out_b, k_new=kn, v_new=vn,
                            scores=st_b, head_ids=hids)
    step_a = ops.StepState(cos, sin) if dyn else None
    step_b = ops.StepState(cos, sin) if dyn else None
    if dyn:
        step_a.set(P, P - 1)
        step_b.set(P, P - 1)

    def per_layer(n):
        if dyn:
            step_a.advance()
        for l in range(L):
            ids = None if hids is None else hids[l]
            if ids is not None and ids.numel() == 0:
                continue
            kc, krc, vc = A[l]
            if dyn:
                ops.attn_decode(q[l], kc, krc, vc, cap, cos, sin, 0, k_new=kn[l], v_new=vn[l], scores=st_a[l], out=out_a[l],
                                workspace=ws, head_ids=ids, step=step_a, n_splits=n_splits)
            else:
                ops.attn_decode(q[l], kc, krc, vc, n, cos, sin, n - 1, k_new=kn[l], v_new=vn[l], scores=st_a[l], out=out_a[l],
                                workspace=ws, head_ids=ids, n_splits=n_splits)

    def chained(n):
        if dyn:
            step_b.advance()
            chain(cap, cos, sin, 0, step=step_b, n_splits=n_splits)
        else:
            chain(n, cos, sin, n - 1, n_splits=n_splits)

    g = None
    if graph:
        assert dyn
        chained(P + 1)                      # warm-up outside the capture (the step state advances: re-set below)
        torch.cuda.synchronize()
        for l in range(L):
            for a, b_ in zip(A[l], Bs[l]):
                b_.copy_(a)
            st_b[l].zero_()
        step_b.set(P, P - 1)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chained(0)
    for t in range(steps):
        n = P + t + 1
        for l in range(L):                  # fresh inputs per token, shared by both sides
            q[l].copy_(rnd(B, H, d)); kn[l].copy_(rnd(B, H, d)); vn[l].copy_(rnd(B, H, d))
        per_layer(n)
        if g is not None:
            g.replay()
        else:
            chained(n)
        torch.cuda.synchronize()
        chain.check()
        for l in range(L):
            ids = None if hids is None else hids[l]
            hs = list(range(H)) if ids is None else [int(x) for x in ids.cpu()]
            for h in hs:
                assert torch.equal(out_a[l][:, h * d:(h + 1) * d], out_b[l][:, h * d:(h + 1) * d]), (t, l, h, "out")
                assert torch.equal(st_a[l][:, h, :n], st_b[l][:, h, :n]), (t, l, h, "stash")
                for a, b_, nm in zip(A[l], Bs[l], ("k", "kr", "v")):
                    assert torch.equal(a[:, h, :n], b_[:, h, :n]), (t, l, h, nm)
    return out_b


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_chain_c2_geometry_static_equals_per_layer_launches(dt):
    # C2: 32 heads x 128, ~2081 rows (a split's chunk is ONE tile: the walking-workgroup form), 5 layers, 4 tokens
    _run_pair(5, 1, 32, 128, 2080, 2176, dt, steps=4)


def test_chain_matches_the_oracle_at_c2_geometry():
    """The first layer of a chained token against the numpy oracle (the per-layer launch is pinned to it elsewhere; this closes
    the loop for the chain itself)."""
    from spatten_amd import ops
    dt, B, H, d, P, cap = "bf16", 1, 32, 128, 700, 1024
    tdt = TORCH_DT[dt]
    q, k, v, past = attn_inputs(B, H, H, d, P, 1, dt, 21)
    cos_h, sin_h = orc.rope_table(cap + 8, d, dt)
    cos, sin = dev(cos_h[:, : d // 2], dt), dev(sin_h[:, : d // 2], dt)
    L = 3
    planes = []
    for l in range(L):
        kc = torch.zeros(B, H, cap, d, dtype=tdt, device="cuda")
        vc, krc = torch.zeros_like(kc), torch.zeros_like(kc)
        kc[:, :, :P], vc[:, :, :P] = dev(past[0], dt), dev(past[1], dt)
        ops.build_shadow(kc, krc, 0, P, cos, sin)
        planes.append((kc, krc, vc))
    qs = [dev(q[:, :, 0], dt) for _ in range(L)]
    ks = [dev(k[:, :, 0], dt) for _ in range(L)]
    vs = [dev(v[:, :, 0], dt) for _ in range(L)]
    outs = [torch.zeros(B, H * d, dtype=tdt, device="cuda") for _ in range(L)]
    stash = [torch.zeros(B, H, cap, dtype=tdt, device="cuda") for _ in range(L)]
    chain = ops.DecodeChain(qs, [p[0] for p in planes], [p[1] for p in planes], [p[2] for p in planes], outs, k_new=ks,
                            v_new=vs, scores=stash)
    chain(P + 1, cos, sin, P)
    torch.cuda.synchronize()
    chain.check()
    import numpy as np
    o, st, (kc_ref, vc_ref) = orc.attention_core(q, k, v, past[0], past[1], np.full((B, 1), P), None, dt)
    for l in range(L):
        np.testing.assert_allclose(host(outs[l])[:, None], o, **OUT_TOL[dt])
        assert np.mean(host(stash[l][:, :, :P + 1])[:, :, None] != st) < 0.02
        assert np.array_equal(host(planes[l][0][:, :, :P + 1]), kc_ref) and np.array_equal(host(planes[l][2][:, :, :P + 1]), vc_ref)


def test_chain_device_length_form_under_a_captured_graph():
    # one captured graph of the chained token replayed for 5 tokens, against the per-layer device-length launches
    _run_pair(4, 1, 32, 128, 2050, 2176, "bf16", steps=5, dyn=True, graph=True)


def test_chain_device_length_form_eager():
    _run_pair(3, 1, 32, 128, 2050, 2176, "bf16", steps=3, dyn=True)


def test_chain_with_per_layer_head_lists():
    # head pruning / a head-parallel rank's survivors: the lists shrink layer by layer, one layer launches nothing
    H = 32
    ids = lambda xs: torch.tensor(xs, dtype=torch.int32, device="cuda")
    hids = [None, ids(list(range(0, 32, 1))[:28]), ids([0, 2, 3, 5, 7, 8, 9, 11, 13, 14, 15, 17, 19, 20, 21, 23, 25, 26, 27, 29, 30, 31, 1, 4][:24]),
            ids([]), ids([1, 4, 6, 30]), ids([0, 31])]
    hids[2] = torch.sort(hids[2]).values.to(torch.int32)
    _run_pair(6, 1, H, 128, 2080, 2176, "bf16", steps=3, hids=hids, n_splits=8)


def test_chain_few_heads_per_rank():
    # a head-parallel rank of C3: 4 heads
    _run_pair(6, 1, 4, 128, 2080, 2176, "bf16", steps=3)


def test_chain_c5_geometry_long_chunks():
    # C5: 40 heads, 8192 kept rows: pipelined tiles, two lanes of workgroups per CU
    _run_pair(4, 1, 40, 128, 8190, 8320, "bf16", steps=3)


def test_chain_c5_geometry_device_length():
    _run_pair(3, 1, 40, 128, 8190, 8320, "bf16", steps=2, dyn=True)


def test_chain_unsupported_shapes_are_refused_not_run():
    from spatten_amd import ops
    tdt = torch.bfloat16
    cos, sin = ops.rope_table(300, 64, tdt, "cuda")
    mk = lambda *s: torch.zeros(*s, dtype=tdt, device="cuda")
    L, B, H, d, cap = 2, 1, 4, 64, 256
    ch = ops.DecodeChain([mk(B, H, d) for _ in range(L)], None, [mk(B, H, cap, d) for _ in range(L)],
                         [mk(B, H, cap, d) for _ in range(L)], [mk(B, H * d) for _ in range(L)])
    with pytest.raises(NotImplementedError):
        ch(100, cos, sin, 99)               # head_dim 64 is not instantiated
