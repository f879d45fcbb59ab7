"""PQ decode timing at 8192 rows (developer tool): MSB-only pass vs bf16 keys."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
H, N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 128
torch.manual_seed(0)
NC = 4
K = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
V = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
pl = []
for i in range(NC):
    p_ = ops.PQPlanes(1, H, N, d, dev); ops.pq_pack(K[i], p_, 0, N); pl.append(p_)
q = torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 8, d, dev, dt)
out = torch.empty(1, H * d, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)
def _time(fn, n=20, reps=5):
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
a = _time(lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws))
b = _time(lambda i: ops.attn_decode_pq(q, pl[i % NC], V[i % NC], N, cos, sin, N - 1, 0.0, out=out, workspace=ws))
c = _time(lambda i: ops.attn_decode_pq(q, pl[i % NC], V[i % NC], N, cos, sin, N - 1, 2.0, out=out, workspace=ws))
print(f"H={H} N={N}: bf16 {a:.2f} us  pq msb-only {b:.2f} us  pq refetch-all {c:.2f} us")
