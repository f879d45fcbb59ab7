"""The 8192-token causal prefill launch alone (BASELINE configs[3] geometry: H = 32, d = 128, bf16), for a kernel trace:
20 calls in reference numerics, then 20 with numerics="fast"."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import ops  # noqa: E402
dt, B, H, d, N = torch.bfloat16, 1, 32, 128, 8192
q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=dt) for _ in range(3))
cos, sin = ops.rope_table(N, d, dt, "cuda")
kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
fl = 4 * B * H * d * N * (N + 1) / 2
for name, kw in (("reference numerics", dict(numerics="reference")), ("fast numerics", dict(numerics="fast"))):
    for _ in range(3):
        ops.attn_prefill(q, kr, v, N, cos, sin, 0, causal=True, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attn_prefill(q, kr, v, N, cos, sin, 0, causal=True, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"prefill q = N = {N} causal, {name}: {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s")
