"""Does touching the o_proj weight on a side stream WHILE the (latency-bound) decode launch runs make the projection that
follows faster?  Per layer, under one HIP graph over 32 layers with their own K/V and weights (1 GB each: nothing survives in
the 256 MB Infinity Cache from one replay to the next):
   A   decode ; gemv(W_o)
   B   decode || touch(W_o) ; gemv(W_o)        touch = the gemv itself into a dummy row (streaming loads), or a torch sum
   C   decode || touch(W_qkv of the next layer) ; gemv(W_o) ; gemv(W_qkv)     (the stacked q/k/v weight, 100 MB)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

H, N = 32, 2081
dev, dt, d, L = torch.device("cuda:0"), torch.bfloat16, 128, 32
hid = H * d
K = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
W = [torch.randn(hid, hid, device=dev, dtype=dt) * 0.02 for _ in range(L)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
kn, vn = torch.randn(1, H, d, device=dev, dtype=dt), torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 64, d, dt, dev)
out = torch.empty(1, hid, device=dev, dtype=dt)
y = torch.empty(1, hid, device=dev, dtype=dt)
dummy = torch.empty(1, hid, device=dev, dtype=dt)
dsum = torch.empty((), device=dev, dtype=torch.float32)
st = torch.empty(1, H, N + 64, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)


def decode(l):
    ops.attn_decode(q, K[l], K[l], V[l], N, cos, sin, N - 1, k_new=kn, v_new=vn, out=out, scores=st, workspace=ws)


def run(mode):
    main, side = torch.cuda.Stream(), torch.cuda.Stream()

    def layer(l):
        if mode == "A":
            decode(l)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if mode == "B_gemv":
                    ops.gemv(out, W[l], out=dummy)
                else:
                    torch.sum(W[l].view(-1), dim=(0,), dtype=torch.float32, out=dsum)
            decode(l)
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
        ops.gemv(out, W[l], out=y)

    with torch.cuda.stream(main):
        layer(0)
        main.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            for l in range(L):
                layer(l)
        g.replay()
        main.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            g.replay()
        main.synchronize()
    return (time.perf_counter() - t) / (20 * L) * 1e6


for mode in ("A", "B_gemv", "B_sum", "A"):
    print(f"{mode}: {run(mode):.2f} us per layer")
