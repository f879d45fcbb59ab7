import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops
dt = torch.bfloat16
B, H, d = 1, 32, 128
for N in (256, 512, 768, 1024, 1536):
    q = torch.randn(B, H, N, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
    out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
    def t(fn, n=10, reps=5):
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
    print(f"N={N}: {t(lambda: ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True)):.1f} us")
