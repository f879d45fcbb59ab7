#!/usr/bin/env python3
"""Phase anatomy of the ping-pong prefill kernel (needs a -DSPATTEN_PF_TRACE build of the library; tools/mb/pf_trace.sh).
Prints, per wave of workgroup 0, the average shader cycles of: matrix phase, wait at barrier 1, staging, softmax,
wait at barrier 2 — over tiles 8..23 of a q = N = 8192 causal prefill."""
import ctypes
import os
import sys

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
numerics = "fast" if len(sys.argv) > 1 and sys.argv[1] == "fast" else "reference"      # probe_pf_trace.py [fast]
print(f"numerics = {numerics}")
for _ in range(2):
    ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True, numerics=numerics)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(8, 16, 8).astype("float64")
names = ["matrix", "bar1 wait", "staging", "softmax", "bar2 wait"]
print("wave " + " ".join(f"{n:>10s}" for n in names) + "      tile")
for w in range(8):
    d_ = [(t[w, :, i + 1] - t[w, :, i]).mean() for i in range(5)]
    tile = (t[w, 1:, 0] - t[w, :-1, 0]).mean()
    dma = (t[w, :, 6] - t[w, :, 0]).mean()          # stamp 6: the stage's LDS-DMA pieces issued; 7: the P.V MFMAs issued
    pv_ = (t[w, :, 7] - t[w, :, 6]).mean()
    qk_ = (t[w, :, 1] - t[w, :, 7]).mean()
    print(f"{w:4d} " + " ".join(f"{x:10.0f}" for x in d_) + f" {tile:9.0f}   matrix = dma {dma:5.0f} + pv {pv_:5.0f} + qk {qk_:5.0f}")
