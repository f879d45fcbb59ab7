"""Decode timing over the profiled quantised planes (developer tool): bf16 keys/values vs the r03 planes (4-bit K, bf16 V) vs
the ABI-4 profiles (key MSB bits, value bits) — MSB pass only and refetch-all, per launch under a HIP graph over rotating
copies (> the Infinity Cache).   python tools/mb/pqv_exp.py [H] [N] [n_splits]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = 128
torch.manual_seed(0)
NC = 4
K = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
V = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(NC)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 8, d, dt, dev)
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
print(f"H={H} N={N} S={S or 'auto'}: bf16 K/V {a:.2f} us")
pl = []
for i in range(NC):
    p_ = ops.PQPlanes(1, H, N, d, dev); ops.pq_pack(K[i], p_, 0, N); pl.append(p_)
b = _time(lambda i: ops.attn_decode_pq(q, pl[i % NC], V[i % NC], N, cos, sin, N - 1, 0.0, out=out, workspace=ws))
c = _time(lambda i: ops.attn_decode_pq(q, pl[i % NC], V[i % NC], N, cos, sin, N - 1, 2.0, out=out, workspace=ws))
print(f"  (4+4, bf16 V) msb-only {b:.2f} us  refetch-all {c:.2f} us")
del pl
for kb, vb in ops.PQ_PROFILES:
    pp = []
    for i in range(NC):
        p_ = ops.PQProfilePlanes(1, H, H, N, d, dev, key_bits=kb, value_bits=vb); ops.pq_pack_planes(K[i], V[i], p_, 0, N); pp.append(p_)
    need = torch.zeros(H, dtype=torch.int32, device=dev)
    b = _time(lambda i: ops.attn_decode_pqv(q, pp[i % NC], N, cos, sin, N - 1, 0.0, out=out, need_lsb=need, workspace=ws, n_splits=S))
    c = _time(lambda i: ops.attn_decode_pqv(q, pp[i % NC], N, cos, sin, N - 1, 2.0, out=out, need_lsb=need, workspace=ws, n_splits=S))
    rowb = d * kb // 8 + d * vb // 8 + 8 + 4
    print(f"  ({kb}+4, V{vb}) msb-only {b:.2f} us ({H * N * rowb / b / 1e6:.2f} TB/s)  refetch-all {c:.2f} us")
    del pp
