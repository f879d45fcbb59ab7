"""Decode launch time at B = 4 / 8 (2080-row caches, 32 heads) as a function of the split count (n_splits argument)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

dev, dt, d, L, H, N = torch.device("cuda:0"), torch.bfloat16, 128, 16, 32, 2081
for B in (2, 4, 8):
    K = [torch.randn(B, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
    V = [torch.randn(B, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
    q = torch.randn(B, H, d, device=dev, dtype=dt)
    kn, vn = torch.randn(B, H, d, device=dev, dtype=dt), torch.randn(B, H, d, device=dev, dtype=dt)
    cos, sin = ops.rope_table(N + 64, d, dt, dev)
    out = torch.empty(B, H * d, device=dev, dtype=dt)
    st = torch.empty(B, H, N + 64, device=dev, dtype=dt)
    ws = ops.DecodeWorkspace(B, H, d, dev)

    def t_us(ns):
        side = torch.cuda.Stream()
        fn = lambda l: ops.attn_decode(q, K[l], K[l], V[l], N, cos, sin, N - 1, k_new=kn, v_new=vn, out=out, scores=st, workspace=ws, n_splits=ns)
        with torch.cuda.stream(side):
            fn(0)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for l in range(L):
                    fn(l)
            g.replay()
            side.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                g.replay()
            side.synchronize()
        return (time.perf_counter() - t) / (20 * L) * 1e6

    bytes_ = B * (2 * H * N * d * 2)
    print(f"B={B}: " + "  ".join(f"S={ns or 'auto'}: {t_us(ns):.2f} us ({bytes_ / t_us(ns) / 1e6 / 8:.0%})" for ns in (0, 1, 2, 3, 4, 6, 8)))
    del K, V
