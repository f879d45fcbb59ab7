"""Split count at long rows: the 16-bit decode step and the profiled-plane decode (MSB pass / refetch-all), n_splits forced.
python tools/mb/split_sweep_long.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import ops
dev, dt, d, L = torch.device("cuda:0"), torch.bfloat16, 128, 4
def tm(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for l in range(4 * L): fn(l % L)
        g.replay(); side.synchronize(); t = time.perf_counter()
        for _ in range(10): g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (40 * L) * 1e6
for N in (8192, 16384):
    for H in (30, 32, 40):
        K = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(L)]
        V = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(L)]
        q = torch.randn(1, H, d, device=dev, dtype=dt)
        cos, sin = ops.rope_table(N + 8, d, dt, dev)
        out = torch.empty(1, H * d, device=dev, dtype=dt)
        ws = ops.DecodeWorkspace(1, H, d, dev)
        pl = []
        for i in range(L):
            p_ = ops.PQProfilePlanes(1, H, H, N, d, dev, key_bits=8, value_bits=8); ops.pq_pack_planes(K[i], V[i], p_, 0, N); pl.append(p_)
        need = torch.zeros(H, dtype=torch.int32, device=dev)
        for label, mk in (("bf16", lambda ns: (lambda l: ops.attn_decode(q, None, K[l], V[l], N, cos, sin, N - 1, out=out, workspace=ws, n_splits=ns))),
                          ("pq88 msb", lambda ns: (lambda l: ops.attn_decode_pqv(q, pl[l], N, cos, sin, N - 1, 0.0, out=out, need_lsb=need, workspace=ws, n_splits=ns))),
                          ("pq88 refetch", lambda ns: (lambda l: ops.attn_decode_pqv(q, pl[l], N, cos, sin, N - 1, 2.0, out=out, need_lsb=need, workspace=ws, n_splits=ns)))):
            print(f"N={N} H={H} {label}: " + "  ".join(f"S={ns or 'auto'}: {tm(mk(ns)):.2f}" for ns in (0, 4, 5, 6, 7, 8, 10, 12)), flush=True)
        del K, V, pl
        torch.cuda.empty_cache()
