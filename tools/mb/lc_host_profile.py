"""Host-side cost of ops.prune_layer_cascade (cProfile over 20 calls into pre-allocated planes)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
L, H, d, CTX, START, RECENT, IMP = 32, 32, 128, 4096, 4, 1024, 1020
cap = 2176
K = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
sc = [torch.randn(H, CTX, device=dev, dtype=dt) for _ in range(L)]
cos, sin = ops.rope_table(CTX, d, dt, dev)
hi = CTX - RECENT
keeps = [IMP - (IMP // 2) * l // (L - 1) for l in range(L)]
Kd = [torch.empty(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
Vd = [torch.empty_like(x) for x in Kd]
Krd = [torch.empty_like(x) for x in Kd]
f = lambda: ops.prune_layer_cascade(sc, [None] * L, 0, K, V, [CTX] * L, [hi] * L, keeps, START, [cap] * L, (cos, sin), None, dst=(Kd, Vd, Krd))
for _ in range(3): f()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): f()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000])
