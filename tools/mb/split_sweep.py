"""Decode launch time by head count x split count x cache length (n_splits forced; 0 = the library's own rule): python tools/mb/split_sweep.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import ops
dev, dt, d, L = torch.device("cuda:0"), torch.bfloat16, 128, 16
def run(H, N, splits):
    K = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
    V = [torch.randn(1, H, N + 64, d, device=dev, dtype=dt) for _ in range(L)]
    q = torch.randn(1, H, d, device=dev, dtype=dt)
    kn, vn = torch.randn(1, H, d, device=dev, dtype=dt), torch.randn(1, H, d, device=dev, dtype=dt)
    cos, sin = ops.rope_table(N + 64, d, dt, dev)
    out = torch.empty(1, H * d, device=dev, dtype=dt)
    st = torch.empty(1, H, N + 64, device=dev, dtype=dt)
    ws = ops.DecodeWorkspace(1, H, d, dev)
    res = []
    for ns in splits:
        side = torch.cuda.Stream()
        fn = lambda l: ops.attn_decode(q, K[l], K[l], V[l], N, cos, sin, N - 1, k_new=kn, v_new=vn, out=out, scores=st, workspace=ws, n_splits=ns)
        with torch.cuda.stream(side):
            fn(0); side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for l in range(L): fn(l)
            g.replay(); side.synchronize(); t = time.perf_counter()
            for _ in range(20): g.replay()
            side.synchronize()
        res.append((ns, (time.perf_counter() - t) / (20 * L) * 1e6))
    print(f"H={H} N={N}: " + "  ".join(f"S={ns or 'auto'}: {u:.2f}" for ns, u in res), flush=True)
for N in (2081, 4100, 8200):
    for H in (4, 8, 12, 16, 20, 24, 28, 32, 40):
        run(H, N, (0, 4, 6, 8, 10, 12, 16, 32))
