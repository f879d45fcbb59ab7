"""Decode launch time when a rank holds FEW heads (head-parallel strong scaling: 4 heads per GPU at 8 x Llama-2-7B, 5 at
Llama-2-13B) as a function of the split count:  probe_few_heads.py [H] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

H, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), (int(sys.argv[2]) if len(sys.argv) > 2 else 2081)
dev, dt, d, L = torch.device("cuda:0"), torch.bfloat16, 128, 32
K = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
kn, vn = torch.randn(1, H, d, device=dev, dtype=dt), torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 64, d, dt, dev)
out = torch.empty(1, H * d, device=dev, dtype=dt)
st = torch.empty(1, H, N + 64, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)


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


print(f"H={H} N={N}: " + "  ".join(f"S={ns or 'auto'}: {t_us(ns):.2f} us" for ns in (0, 4, 8, 12, 16, 24, 32, 48, 64)))
