"""Phase stamps of the chained decode launch (library built with -DSPATTEN_CHAIN_TRACE, SPATTEN_LIB=...): per layer, on the
device-wide 100 MHz clock.  python tools/mb/chain_trace.py [heads] [rows] [layers]"""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import _lib, ops  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2081
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B, d, dt, dev = 1, 128, torch.bfloat16, "cuda"
cap = (N + 64 + 127) // 128 * 128
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
out = [torch.zeros(B, H * d, dtype=dt, device=dev) for _ in range(L)]
st = [torch.zeros(B, H, cap, dtype=dt, device=dev) for _ in range(L)]
chain = ops.DecodeChain(q, K, Kr, V, out, k_new=kn, v_new=vn, scores=st)
lib = _lib.load()
S = lib.spatten_decode_auto_splits(B, H, d, N)
nwg = S * H * B * (2 if N > 8 * 320 else 1)
buf = torch.zeros(L * nwg * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    chain(N, cos, sin, N - 1)
torch.cuda.synchronize()
lib.spatten_debug_set_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.spatten_debug_set_chain_trace(buf.data_ptr()) == 0
chain(N, cos, sin, N - 1)
torch.cuda.synchronize()
lib.spatten_debug_set_chain_trace(None)
t = buf.cpu().numpy().reshape(L, nwg, 16).astype(np.float64) * 0.01      # us
names = ["step start", "flag seen (wave 0)", "q staged (barrier passed)", "q rotated, next keys requested", "tile done",
         "workgroup reduced", "merger: partials landed", "step end (flag stored)", "non-merger: granules issued",
         "non-merger: end barrier passed", "loop top (this layer)", "layer table entry read"]
order = [10, 11, 0, 1, 2, 3, 4, 5, 8, 9, 6, 7]
t0 = t[t > 0].min()
print(f"H={H} N={N} L={L} S={S}: layer period (flag-stored of the last unit, layer to layer):")
ends = np.array([t[l, :, 7].max() for l in range(L)])
print("  ", np.round(np.diff(ends), 2))
for l in (L // 2, L - 2):
    base = t[l - 1, :, 7].max()           # the previous layer's last completion word
    print(f"layer {l}: times relative to the previous layer's last completion (us): min / median / max over workgroups")
    for s in order:
        nm = names[s]
        x = t[l, :, s]
        x = x[x > 0] - base
        if x.size:
            print(f"  {nm:34s} n={x.size:4d}  {x.min():7.2f} {np.median(x):7.2f} {x.max():7.2f}")
