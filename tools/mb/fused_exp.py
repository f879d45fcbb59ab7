"""The layer-step at Llama-2-7B geometry under ONE HIP graph over 32 layers with their own weights and K/V (nothing survives in
the Infinity Cache from one replay to the next):
   A   spatten_gemv(stacked q/k/v) ; plain decode step ; spatten_gemv(o_proj)           (r03: 97 launches per token)
   B   fused projection + attention launch (decode_qkv_kernel) ; spatten_gemv(o_proj)    (65 launches per token)
   C   the whole attention module in one launch (o_proj inside; SPATTEN_FUSED_OPROJ=0 -> B)  (33 launches per token)
python tools/mb/fused_exp.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
H, d, L = 32, 128, 32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2081
dev, dt = torch.device("cuda:0"), torch.bfloat16
hid = H * d
cap = (N + 64 + 127) // 128 * 128
torch.manual_seed(0)
K = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
KR = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
Wqkv = [torch.randn(3 * hid, hid, device=dev, dtype=dt) * hid ** -0.5 for _ in range(L)]
Wo = [torch.randn(hid, hid, device=dev, dtype=dt) * hid ** -0.5 for _ in range(L)]
cos, sin = ops.rope_table(cap + 8, d, dt, dev)
x = torch.randn(1, 1, hid, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)
st = torch.zeros(1, H, cap, device=dev, dtype=dt)
out = torch.empty(1, hid, device=dev, dtype=dt)
y = torch.empty(1, 1, hid, device=dev, dtype=dt)


def token_a():
    h = x
    for i in range(L):
        qkv = ops.gemv(h, Wqkv[i]).view(3, H, d)
        o = ops.attn_decode(qkv[0][None], K[i], KR[i], V[i], N, cos, sin, N - 1, k_new=qkv[1][None], v_new=qkv[2][None], scores=st,
                            out=out, workspace=ws, layout=cap)
        h = ops.gemv(o.view(1, 1, hid), Wo[i], out=y)


def token_b():
    h = x
    for i in range(L):
        o = ops.attn_decode_qkv(h, Wqkv[i], None, H, K[i], KR[i], V[i], N, cos, sin, N - 1, scores=st, out=out, workspace=ws, layout=cap)
        h = ops.gemv(o.view(1, 1, hid), Wo[i], out=y)


def token_c():
    h = x
    for i in range(L):
        o, h = ops.attn_decode_qkv(h, Wqkv[i], None, H, K[i], KR[i], V[i], N, cos, sin, N - 1, scores=st, out=out, workspace=ws,
                                   layout=cap, proj=(Wo[i], None, y.view(1, hid)))
        h = h.view(1, 1, hid)


def _time(fn, reps=20):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        for _ in range(3): g.replay()
        side.synchronize(); t = time.perf_counter()
        for _ in range(reps): g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


for rnd in range(2):
    a, b, c = _time(token_a), _time(token_b), _time(token_c)
    print(f"rows {N}: separate {a:.1f} us/token ({a / L:.2f} us/layer, {1e6 / a:.0f} tok/s)   fused qkv {b:.1f} us/token ({b / L:.2f} us/layer, {1e6 / b:.0f} tok/s)"
          f"   whole module {c:.1f} us/token ({c / L:.2f} us/layer, {1e6 / c:.0f} tok/s)")
