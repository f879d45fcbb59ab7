"""Kernel-level view of the q = N = 2048 prefill (developer tool): run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dt = torch.bfloat16
B, H, d = 1, 32, 128
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
q = torch.randn(B, H, N, d, device="cuda", dtype=dt)
k = torch.randn(B, H, N, d, device="cuda", dtype=dt)
v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
cos, sin = ops.rope_table(N, d, dt, "cuda")
kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
for _ in range(30):
    ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True)
e1.record(); torch.cuda.synchronize()
print(f"N={N}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call")
