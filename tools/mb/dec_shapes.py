"""The plain decode step at several shapes (B = 1, bf16, d = 128), one HIP graph over rotating layers, us per layer:
   python tools/mb/dec_shapes.py        (SPATTEN_DECODE_TEAM=256|512, SPATTEN_DECODE_TEAM_PIPE=0|1 select the team)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dt, d = torch.bfloat16, 128
out_line = []
SHAPES = ((32, 2081, 32), (32, 4096, 24), (32, 8192, 12), (40, 16384, 8), (40, 8192, 12))
if os.environ.get("DEC_SHAPES") == "tail":      # how much the nearly empty last row group of a 2049..2112-row step costs
    SHAPES = ((32, 2040, 32), (32, 2048, 32), (32, 2049, 32), (32, 2081, 32), (32, 2112, 32), (32, 2176, 32))
if os.environ.get("DEC_N"):
    SHAPES = tuple((32, int(n), 32) for n in os.environ["DEC_N"].split(","))
for H, N, L in SHAPES:
    cap = N + 64
    KR = [torch.randn(1, H, cap, d, device="cuda", dtype=dt) for _ in range(L)]
    V = [torch.randn(1, H, cap, d, device="cuda", dtype=dt) for _ in range(L)]
    q = torch.randn(1, H, d, device="cuda", dtype=dt)
    kn, vn = torch.randn(1, H, d, device="cuda", dtype=dt), torch.randn(1, H, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(cap, d, dt, "cuda")
    st = torch.empty(1, H, cap, device="cuda", dtype=dt)
    out = torch.empty(1, H * d, device="cuda", dtype=dt)
    K = [torch.empty_like(x) for x in KR]
    def step():
        for i in range(L):
            ops.attn_decode(q, K[i], KR[i], V[i], N, cos, sin, N - 1, k_new=kn, v_new=vn, scores=st, out=out)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step(); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
        for _ in range(3): g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(10): g.replay()
        e1.record(side); side.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10 / L
    by = 2 * H * N * d * 2 + H * N * 2
    out_line.append(f"H={H} N={N}: {us:.2f} us ({by / us / 1e6 / 8:.3f} of peak)")
    del KR, V, K
print(" | ".join(out_line))
