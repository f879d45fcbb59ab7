"""Round 6: the chained decode launch against the per-layer launches at the C2 geometry (and variants) — same planes, same
inputs, HIP-graph replays timed with events.  python tools/mb/chain_bench.py [heads] [rows] [layers]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import ops  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2081
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
depth = int(os.environ.get("CHAIN_DEPTH", "0"))
NS = int(os.environ.get("CHAIN_SPLITS", "0"))      # n_splits of BOTH forms (0 = each form's default)
LAY = int(os.environ.get("CHAIN_LAYOUT", "0"))      # kv_len_layout of BOTH forms (0 = equal chunks)
B, d, dt = 1, 128, torch.bfloat16
cap = (N + 64 + 127) // 128 * 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32, generator=g).to(dt)
cos, sin = ops.rope_table(cap + 8, d, dt, dev)
K, Kr, V = [], [], []
for l in range(L):
    k = torch.zeros(B, H, cap, d, dtype=dt, device=dev); k[:, :, :N] = rnd(B, H, N, d)
    v = torch.zeros_like(k); v[:, :, :N] = rnd(B, H, N, d)
    kr = torch.zeros_like(k)
    ops.build_shadow(k, kr, 0, N - 1, cos, sin)
    K.append(k); Kr.append(kr); V.append(v)
q = [rnd(B, H, d) for _ in range(L)]
kn = [rnd(B, H, d) for _ in range(L)]
vn = [rnd(B, H, d) for _ in range(L)]
out_a = [torch.zeros(B, H * d, dtype=dt, device=dev) for _ in range(L)]
out_b = [torch.zeros_like(x) for x in out_a]
st = [torch.zeros(B, H, cap, dtype=dt, device=dev) for _ in range(L)]
st_b = [torch.zeros_like(x) for x in st]
ws = ops.DecodeWorkspace(B, H, d, dev)
chain = ops.DecodeChain(q, K, Kr, V, out_b, k_new=kn, v_new=vn, scores=st_b, depth=depth)


def per_layer():
    for l in range(L):
        ops.attn_decode(q[l], K[l], Kr[l], V[l], N, cos, sin, N - 1, k_new=kn[l], v_new=vn[l], scores=st[l], out=out_a[l], workspace=ws, layout=LAY, n_splits=NS)


def chained():
    chain(N, cos, sin, N - 1, layout=LAY, n_splits=NS)


def graph_of(fn):
    fn(); fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    return gr


def time_graph(gr, reps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ga, gb = graph_of(per_layer), graph_of(chained)
chain.check()
same = all(torch.equal(a, b) for a, b in zip(out_a, out_b)) and all(torch.equal(a[:, :, :N], b[:, :, :N]) for a, b in zip(st, st_b))
res = []
for rep in range(3):
    ta, tb = time_graph(ga), time_graph(gb)
    res.append((ta, tb))
chain.check()
bytes_layer = 2 * B * H * N * d * 2 + 2 * B * H * d * 2 + B * H * N * 2
for ta, tb in res:
    print(f"H={H} N={N} L={L} layout={LAY} n_splits={NS}: per-layer {ta:8.1f} us/token = {ta / L:6.2f} us/layer ({bytes_layer / (ta / L) / 1e6:.2f} TB/s) | "
          f"chained {tb:8.1f} us/token = {tb / L:6.2f} us/layer ({bytes_layer / (tb / L) / 1e6:.2f} TB/s) | x{ta / tb:.3f} | bit-identical {same}")
