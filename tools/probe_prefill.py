"""Timing probe for the prefill flash kernel (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatten_amd import ops
dt = torch.bfloat16
B, H, d = 1, 32, 128
for N in (2048, 8192):
    q = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    k = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N, d, dt, "cuda")
    kr = ops.rope_single(k, cos, sin)
    out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
    for name, kw in (("causal", dict(causal=True)), ("causal+colimp", dict(causal=True, col_importance=torch.zeros(B, H, N, device="cuda")))):
        for _ in range(2):
            ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 4 * B * H * d * N * (N + 1) / 2
        print(f"prefill N={N} {name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (causal flops)")
