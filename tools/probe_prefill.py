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
    for name, kw in (("causal", dict(causal=True, numerics="reference")), ("causal fast", dict(causal=True, numerics="fast")),
                     ("causal+colimp", dict(causal=True, col_importance=torch.zeros(B, H, N, device="cuda")))):
        reps = 20 if name.startswith("causal") and "colimp" not in name else 5
        for _ in range(3):
            ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 4 * B * H * d * N * (N + 1) / 2
        print(f"prefill N={N} {name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (causal flops)")
# reference-parity mode: full stash [B,H,q,N] written (modify_llama.py:116-119)
N = 4096
q = torch.randn(B, H, N, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, N, H * d, device="cuda", dtype=dt); st = torch.empty(B, H, N, N, device="cuda", dtype=dt)
for _ in range(2): ops.attn_prefill(q, kr, v, N, cos, sin, 0, causal=True, out=out, scores=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): ops.attn_prefill(q, kr, v, N, cos, sin, 0, causal=True, out=out, scores=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"prefill N={N} causal+stash: {ms:.3f} ms  (stash {st.numel()*2/2**30:.2f} GiB -> {st.numel()*2/ms/1e9:.2f} TB/s of stash writes)")
# explicit HF-style additive causal mask [B,1,q,N] (what transformers 4.33 passes; the plugin reads it unless assume_causal)
N = 4096
q = torch.randn(B, H, N, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
i = torch.arange(N, device="cuda")
mask = torch.where(i[None, :] <= i[:, None], 0.0, torch.finfo(dt).min).to(dt)[None].contiguous()
for name, kw in (("mask, no stash", dict(mask=mask)), ("causal flag, no stash", dict(causal=True))):
    for _ in range(2): ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"prefill N={N} {name}: {e0.elapsed_time(e1) / 5:.3f} ms")
