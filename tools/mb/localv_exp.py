"""Local V pruning at long context: time the pieces (scores-only pass, select, kept-row P.V) against the plain fused
decode.  Usage: python tools/mb/localv_exp.py [N] [H] [keep_frac]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops, cascade

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
d, dev, dt = 128, torch.device("cuda:0"), torch.bfloat16
torch.manual_seed(0)
NC = 3
K = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
V = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 8, d, dev, dt)
out = torch.empty(1, H * d, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)
stash = torch.empty(1, H, N, device=dev, dtype=dt)
lse = torch.empty(1, H, 2, device=dev, dtype=torch.float32)
keep = int(N * frac)


def _time(fn, n=10, reps=5):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(n):
                fn(i)
        for _ in range(2):
            g.replay()
        side.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


r = {}
r["plain"] = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws))
r["scores_only"] = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, scores=stash, lse=lse, scores_only=True, workspace=ws))
ops.attn_decode(q, None, K[0], V[0], N, cos, sin, N - 1, scores=stash, lse=lse, scores_only=True, workspace=ws)
idx = ops.topk_select(stash.reshape(H, N), 0, N, keep)
r["select"] = _time(lambda i: ops.topk_select(stash.reshape(H, N), 0, N, keep))
r["pv"] = _time(lambda i: ops.pv_gather(stash, lse, V[i % NC], idx, out=out))
r["local_v"] = _time(lambda i: cascade.local_v_decode(q, K[i % NC], V[i % NC], N, cos, sin, N - 1, keep, workspace=ws, out=out, stash=stash, lse=lse))
print(f"N={N} H={H} keep={keep}: " + " ".join(f"{k}={v:.1f}us" for k, v in r.items()))
