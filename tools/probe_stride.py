#!/usr/bin/env python3
"""Decode time vs the per-head pitch of the KV slab (rows of capacity): power-of-two pitches camp on HBM channels.
    python tools/probe_stride.py [H] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d, dt, dev = 128, torch.bfloat16, torch.device("cuda", 0)
cos, sin = ops.rope_table(N + 64, d, dt, dev)
q = torch.randn(1, H, d, device=dev).to(dt)
out = torch.empty(1, H * d, dtype=dt, device=dev)
ws = ops.DecodeWorkspace(1, H, d, dev)


def timed(fn, n=8, reps=5):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for i in range(n):
                fn(i)
        gr.replay(); side.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


for pad in (0, 64, 128, 192, 256, 384, 512, 1024, 1152):
    cap = N + pad
    nb = max(2, int(600e6 // (H * cap * d * 4)) + 1)          # rotate over > 600 MB
    K = [torch.randn(1, H, cap, d, device=dev).to(dt) for _ in range(nb)]
    V = [torch.randn(1, H, cap, d, device=dev).to(dt) for _ in range(nb)]
    t = timed(lambda i: ops.attn_decode(q, None, K[i % nb], V[i % nb], N, cos, sin, N - 1, out=out, workspace=ws))
    print(f"H={H} N={N} cap={cap:6d} pitch={cap * d * 2 / 1024:8.1f} KiB  {t:7.2f} us  {H * N * d * 4 / t / 1e6:5.2f} TB/s")
    del K, V
