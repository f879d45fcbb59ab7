"""ONE prefill shape, for per-shape rocprofv3 kernel stats:  [PF_HEADS=h] probe_prefill_shape.py q_len kv_len [fast]
(causal flash leg, Llama-2-7B geometry, bf16; kv_len > q_len = a block appended to a cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

ql, N = int(sys.argv[1]), int(sys.argv[2])
numerics = "fast" if len(sys.argv) > 3 and sys.argv[3] == "fast" else "reference"
dt, B, H, d = torch.bfloat16, 1, int(os.environ.get("PF_HEADS", "32")), 128
q = torch.randn(B, H, ql, d, device="cuda", dtype=dt)
k = torch.randn(B, H, N, d, device="cuda", dtype=dt)
v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
cos, sin = ops.rope_table(N, d, dt, "cuda")
kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, ql, H * d, device="cuda", dtype=dt)
run = lambda: ops.attn_prefill(q, kr, v, N, cos, sin, N - ql, causal=True, out=out, numerics=numerics)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
P = N - ql
fl = 4 * B * H * d * (ql * P + ql * (ql + 1) / 2)
print(f"prefill q={ql} N={N} causal numerics={numerics}: {ms * 1e3:.1f} us per call (V transpose + flash [+ merge])  {fl / ms / 1e9:.1f} TFLOP/s")
