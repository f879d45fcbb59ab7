#!/usr/bin/env python3
"""Local V pruning (scores-only decode -> per-head top-k of the logits -> P·V over the kept V rows) vs the plain fused
decode, device time per layer.   python tools/probe_localv.py [N] [keep_fraction]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402
from spatten_amd.cascade import local_v_decode  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
H, d, dt, dev = 32, 128, torch.bfloat16, torch.device("cuda", 0)
cos, sin = ops.rope_table(N + 64, d, dt, dev)
q = torch.randn(1, H, d, device=dev).to(dt)
NB = 4
Kr = [torch.randn(1, H, N, d, device=dev).to(dt) for _ in range(NB)]
V = [torch.randn(1, H, N, d, device=dev).to(dt) for _ in range(NB)]
out = torch.empty(1, H * d, dtype=dt, device=dev)
ws = ops.DecodeWorkspace(1, H, d, dev)


def timed(fn, n=8, reps=5):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(n):
            fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


vk = max(int(round(keep * N)), 1)
print(f"N={N} keep {vk} V rows per head")
print(f"plain decode          : {timed(lambda i: ops.attn_decode(q, None, Kr[i % NB], V[i % NB], N, cos, sin, N - 1, out=out, workspace=ws)):8.1f} us (host-timed)")
print(f"local-V pruned decode : {timed(lambda i: local_v_decode(q, Kr[i % NB], V[i % NB], N, cos, sin, N - 1, vk)):8.1f} us (host-timed, 3 launches + allocations)")
