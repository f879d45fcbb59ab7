"""Decode step with the fused cascade-importance accumulation / head importance on and off (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
H, N, d = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2081, 128
torch.manual_seed(0)
NC = 8
cap = N + 64
K = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(NC)]
V = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(NC)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(cap + 8, d, dev, dt)
out = torch.empty(1, H * d, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)
stash = [torch.zeros(1, H, cap, device=dev, dtype=dt) for _ in range(2)]
lse = [torch.zeros(1, H, 2, device=dev, dtype=torch.float32) for _ in range(2)]
lse[0][..., 1] = 1; lse[1][..., 1] = 1
acc = torch.zeros(H, cap, device=dev, dtype=torch.float32)
habs = torch.zeros(H, device=dev, dtype=torch.float32)
def _time(fn, n=32, reps=5):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(n): fn(i)
        for _ in range(2): g.replay()
        side.synchronize(); t = time.perf_counter()
        for _ in range(reps): g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6
a = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws, scores=stash[i & 1], lse=lse[i & 1]))
b = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws, scores=stash[i & 1], lse=lse[i & 1],
                                    cascade=(acc, stash[(i & 1) ^ 1], lse[(i & 1) ^ 1], N - 1)))
c = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws, scores=stash[i & 1], lse=lse[i & 1], head_abs=habs))
print(f"N={N}: stash+lse {a:.2f} us | + cascade accumulation {b:.2f} us | + head importance {c:.2f} us")
