#!/usr/bin/env python3
"""Phase anatomy of the one-wave-per-SIMD prefill kernel (needs a -DSPATTEN_PF_TRACE build: tools/mb/w4_trace.sh).
Per wave of workgroup 0, average shader cycles over tiles 40..55 of a q = N = 8192 causal prefill of: wait + barrier,
DMA issue, half A (32 MFMAs), half B (32 MFMAs), and the whole tile."""
import ctypes
import os
import sys

os.environ["SPATTEN_PREFILL_W4"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import _lib, ops  # noqa: E402

lib = _lib.load()
buf = torch.zeros(8 * 16 * 8, dtype=torch.int64, device="cuda")
lib.spatten_debug_set_pf_trace.argtypes = [ctypes.c_void_p]
assert lib.spatten_debug_set_pf_trace(buf.data_ptr()) == 0
dt, B, H, d, N = torch.bfloat16, 1, 32, 128, 8192
q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=dt) for _ in range(3))
cos, sin = ops.rope_table(N, d, dt, "cuda")
kr = ops.rope_single(k, cos, sin)
out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
kw = dict(numerics="fast") if len(sys.argv) > 1 and sys.argv[1] == "fast" else {}
for _ in range(2):
    ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"whole call {ms * 1e3:.1f} us = {4 * B * H * d * N * (N + 1) / 2 / ms / 1e9:.0f} TFLOP/s; per workgroup-tile {ms * 1e6 / (32 * sum(4 * b + 5 for b in range(32)) / 256):.0f} ns")
t = buf.cpu().numpy().reshape(8, 16, 8).astype("float64")
names = ["wait+bar", "dma issue", "half A", "half B"]
print("wave " + " ".join(f"{n:>10s}" for n in names) + "      tile")
for w in range(4):
    d_ = [(t[w, :, i + 1] - t[w, :, i]).mean() for i in range(4)]
    tile = (t[w, 1:, 0] - t[w, :-1, 0]).mean()
    print(f"{w:4d} " + " ".join(f"{x:10.0f}" for x in d_) + f" {tile:9.0f}")
