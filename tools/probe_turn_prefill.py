"""Turn prefill of the multi-turn protocol (developer tool): q_len new tokens on top of a 2048-row pruned cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops
dt = torch.bfloat16
B, H, d, P = 1, 32, 128, 2048
for ql in (16, 64, 128, 256, 512, 1024):
    N = P + ql
    q = torch.randn(B, H, ql, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
    out = torch.empty(B, ql, H * d, device="cuda", dtype=dt)
    def t(fn, n=10, reps=5):       # device time: n calls captured into one HIP graph (no host launch gaps)
        import time
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn(); side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(n): fn()
            for _ in range(2): g.replay()
            side.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): g.replay()
            side.synchronize()
        return (time.perf_counter() - t0) / (n * reps) * 1e6
    a = t(lambda: ops.attn_prefill(q, kr, v, N, cos, sin, P, out=out, causal=True))
    fl = 4 * B * H * d * (ql * P + ql * (ql + 1) / 2)
    print(f"P={P} q_len={ql}: {a:.1f} us per layer ({fl / a / 1e6:.0f} TFLOP/s)")
