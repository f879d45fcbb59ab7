"""C4 (BASELINE.json configs[3]): prefill q = N = 8192 over progressively quantised keys vs bf16 keys (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops
dt = torch.bfloat16
B, H, d = 1, 32, 128
for N in (4096, 8192):
    q = torch.randn(B, H, N, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
    planes = ops.PQPlanes(B, H, N, d, "cuda"); ops.pq_pack(kr, planes, 0, N)
    out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
    need = torch.empty(B, H, N, dtype=torch.int32, device="cuda")
    def t(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
    a = t(lambda: ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True))
    res = []
    for thr in (0.0, 0.05, 2.0):       # nobody refetches / the traces' threshold / everybody refetches
        ms = t(lambda: ops.attn_prefill_pq(q, planes, v, N, cos, sin, 0, thr, causal=True, out=out, need_lsb=need))
        res.append(f"thr {thr}: {ms:.3f} ms ({float(need.float().mean()) * 100:.0f} % rows refetched)")
    print(f"N={N}: bf16 keys {a:.3f} ms | PQ keys " + " | ".join(res))
