"""Sweep decode kernel tuning knobs (UNR via env, splits) — developer tool."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from spatten_amd import ops
dt = torch.bfloat16
B, H, d, L = int(os.environ.get("PB", "1")), 32, 128, 32
res = []
for N in (2048, 4096):
    caches = [(torch.randn(B, H, N + 64, d, device="cuda", dtype=dt), torch.randn(B, H, N + 64, d, device="cuda", dtype=dt)) for _ in range(L)]
    q = torch.randn(B, H, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N + 64, d, dt, "cuda")
    scores = torch.empty(B, H, N + 64, device="cuda", dtype=dt)
    out = torch.empty(B, H * d, device="cuda", dtype=dt)
    for ns in [int(x) for x in os.environ.get("PSPLITS", "0").split(",")]:
        def step():
            for kc, vc in caches:
                ops.attn_decode(q, None, kc, vc, N, cos, sin, N - 1, out=out, scores=scores, n_splits=ns)
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 / L * 1e3
        res.append(f"N={N} ns={ns}: {t:.2f}us {2*B*H*N*d*2/t/1e6:.2f}TB/s")
    del caches
print(" | ".join(res))
''' % ROOT
for unr in (1, 2, 4):
    env = dict(os.environ, SPATTEN_DECODE_UNR=str(unr), PSPLITS=os.environ.get("PSPLITS", "4,8,16,32,64"))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"UNR={unr}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
