"""How fast is the decode launch when its K / V rows already sit in the Infinity Cache (the same layer replayed) against
rotating over 32 layers (1.1 GB: every launch streams from HBM)?  Upper bound of what ANY prefetch scheme could buy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatten_amd import ops
dev, dt = torch.device("cuda"), torch.bfloat16
L, d = 32, 128
for H, n, cap in ((32, 2080, 2176), (32, 8192, 8256), (4, 2080, 2176)):
    cos, sin = ops.rope_table(cap, d, dt, dev)
    Kr = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
    V = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
    q = torch.randn(1, H, d, device=dev, dtype=dt)
    out = torch.empty(1, H * d, device=dev, dtype=dt)
    st = torch.empty(1, H, cap, device=dev, dtype=dt)
    ws = ops.DecodeWorkspace(1, H, d, dev)
    for name, pick in (("rotating over 32 layers (HBM)", lambda l: l), ("same layer (cache-warm)", lambda l: 0), ("two layers alternating", lambda l: l & 1)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            f = lambda: [ops.attn_decode(q, None, Kr[pick(l)], V[pick(l)], n, cos, sin, n - 1, scores=st, out=out, workspace=ws) for l in range(L)]
            f(); s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                f()
            for _ in range(5): g.replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(40): g.replay()
            e1.record(s); s.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40 / L
        print(f"H={H} n={n}: {name:34s} {us:7.2f} us per launch  ({2 * H * n * d * 2 / us / 1e6:.2f} TB/s)", flush=True)
    del Kr, V
