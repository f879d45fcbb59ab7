import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops
dt = torch.bfloat16
B, H, d = 1, 32, 128
for N in (4096, 8192):
    q = torch.randn(B, H, N, d, device="cuda", dtype=dt); k = torch.randn(B, H, N, d, device="cuda", dtype=dt); v = torch.randn(B, H, N, d, device="cuda", dtype=dt)
    cos, sin = ops.rope_table(N, d, dt, "cuda"); kr = ops.rope_single(k, cos, sin)
    out = torch.empty(B, N, H * d, device="cuda", dtype=dt)
    lse = torch.empty(B, H, N, 2, dtype=torch.float32, device="cuda")
    acc = torch.zeros(H, N, dtype=torch.float32, device="cuda")
    def t(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
    a = t(lambda: ops.attn_prefill(q, kr, v, N, cos, sin, 0, out=out, causal=True, lse=lse))
    b = t(lambda: ops.importance_accumulate_prefill(acc, q, kr, N, cos, sin, 0, lse, causal=True))
    print(f"N={N}: flash+lse {a:.3f} ms, colprob (cascade importance) {b:.3f} ms")
