#!/usr/bin/env python3
"""Device time of the progressive-quant decode (graph-captured, no host launch cost) next to the bf16-key decode.
    python tools/probe_pq.py [N] [H]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d, dt, dev = 128, torch.bfloat16, torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
cos, sin = ops.rope_table(N + 64, d, dt, dev)
Kr, V, q = rnd(1, H, N, d), rnd(1, H, N, d), rnd(1, H, d)
planes = ops.PQPlanes(1, H, N, d, dev)
ops.pq_pack(Kr, planes, 0, N)
out = torch.empty(1, H * d, dtype=dt, device=dev)
ws = ops.DecodeWorkspace(1, H, d, dev)


def timed(fn, n=20, reps=5):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(n):
                fn()
        gr.replay(); side.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


rows = H * N
print(f"N={N} H={H}")
t = timed(lambda: ops.attn_decode(q, None, Kr, V, N, cos, sin, N - 1, out=out, workspace=ws))
print(f"bf16 keys      : {t:7.2f} us  {rows * 512 / t / 1e6:6.2f} TB/s")
t = timed(lambda: ops.attn_decode_pq(q, planes, V, N, cos, sin, N - 1, 0.0, out=out, workspace=ws))
print(f"pq msb only    : {t:7.2f} us  {rows * 324 / t / 1e6:6.2f} TB/s (incl. the skipped refetch launch)")
t = timed(lambda: ops.attn_decode_pq(q, planes, V, N, cos, sin, N - 1, 2.0, out=out, workspace=ws))
print(f"pq refetch all : {t:7.2f} us  {rows * (324 + 388) / t / 1e6:6.2f} TB/s")
