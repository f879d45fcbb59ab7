"""Grouped-query decode step: the matrix-core form (decode_gqa.hip) against one workgroup column per query head, same planes,
graph replays rotating over enough layer planes to defeat the Infinity Cache.  usage: gqa_bench.py [H Hkv N [B]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spatten_amd import ops

H, Hkv, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 8, 16384)))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
NS = int(os.environ.get("GQA_NS", "0"))          # forced split count (0 = the launch's own choice)
MODES = tuple(int(x) for x in os.environ.get("GQA_MODES", "0,1,0,1").split(","))
d, tdt = 128, torch.bfloat16
cap = N + 64
plane = B * Hkv * cap * d * 2
L = max(4, int(600e6 // (2 * plane)) + 1)
cos, sin = ops.rope_table(cap + 8, d, tdt, "cuda")
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(tdt)
planes = []
for l in range(L):
    kr, v = rnd(B, Hkv, cap, d), rnd(B, Hkv, cap, d)
    planes.append((torch.zeros_like(kr), kr, v))
q, kn, vn = rnd(B, H, d), rnd(B, Hkv, d), rnd(B, Hkv, d)
out = torch.zeros(B, H * d, dtype=tdt, device="cuda")
st = torch.zeros(B, H, cap, dtype=tdt, device="cuda")
ws = ops.DecodeWorkspace(B, H, d, "cuda")
res = {}
for mode in MODES:
    ops.set_decode_gqa(mode)

    def token():
        for kc, krc, vc in planes:
            ops.attn_decode(q, kc, krc, vc, N, cos, sin, N - 1, k_new=kn, v_new=vn, scores=st, out=out, workspace=ws, n_splits=NS)

    token()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        token()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * L)
    uniq = B * Hkv * N * d * 2 * 2
    res.setdefault(mode, []).append(us)
    print(f"H={H} Hkv={Hkv} N={N} B={B} layers={L} ns={NS} mode={mode}: {us:7.2f} us/step  unique K/V {uniq / us / 1e6:6.2f} TB/s "
          f"({uniq / us / 1e6 / 8.0:.3f} of peak)", flush=True)
    ws.check()
