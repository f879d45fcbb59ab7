"""Phase stamps of the grouped-query decode step (library built with -DSPATTEN_GQA_TRACE, SPATTEN_LIB=...), device-wide 100 MHz
clock.  python tools/mb/gqa_trace.py [H Hkv N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spatten_amd import _lib, ops  # noqa: E402

H, Hkv, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 8, 16384)))
B, d, tdt = 1, 128, torch.bfloat16
cap = N + 64
L = max(4, int(600e6 // (2 * B * Hkv * cap * d * 2)) + 1)
cos, sin = ops.rope_table(cap + 8, d, tdt, "cuda")
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(tdt)
planes = [(None, rnd(B, Hkv, cap, d), rnd(B, Hkv, cap, d)) for _ in range(L)]
planes = [(torch.zeros_like(kr), kr, v) for _, kr, v in planes]
q, kn, vn = rnd(B, H, d), rnd(B, Hkv, d), rnd(B, Hkv, d)
out = torch.zeros(B, H * d, dtype=tdt, device="cuda")
st = torch.zeros(B, H, cap, dtype=tdt, device="cuda")
ws = ops.DecodeWorkspace(B, H, d, "cuda")
ops.set_decode_gqa(1)
lib = _lib.load()
lib.spatten_debug_set_gqa_trace.argtypes = [ctypes.c_void_p]
nwg = 64 * Hkv * B
buf = torch.zeros(nwg * 4 * 48, dtype=torch.int64, device="cuda")


def token(n=L):
    for kc, krc, vc in planes[:n]:
        ops.attn_decode(q, kc, krc, vc, N, cos, sin, N - 1, k_new=kn, v_new=vn, scores=st, out=out, workspace=ws)


token()
torch.cuda.synchronize()
assert lib.spatten_debug_set_gqa_trace(buf.data_ptr()) == 0
token(3)            # the stamps of the LAST launch survive (back-to-back launches: a warm pipeline)
torch.cuda.synchronize()
lib.spatten_debug_set_gqa_trace(None)
t = buf.cpu().numpy().reshape(-1, 4, 48).astype(np.float64) * 0.01
live = t[:, 0, 0] > 0
t = t[live]
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
names = ["wave start", "two tiles requested", "queries rotated", "tile 0 keys landed", "tile 0 consumed", "tile 1 keys landed",
         "tile 0 refill requested", "stream done", "partials in LDS", "partial published", "head's partials landed", "merged head stored"]
print(f"H={H} Hkv={Hkv} N={N}: {t.shape[0]} workgroups; us after the first wave start: min / median / max over (workgroup, wave)")
for s_, nm in enumerate(names):
    x = t[:, :, s_]
    x = x[x > 0] - t0
    if x.size:
        print(f"  {nm:28s} {x.min():7.2f} {np.median(x):7.2f} {x.max():7.2f}   (n={x.size})")

it_names = ["keys ready", "S done", "key refill issued", "softmax done", "values ready", "P.V done", "value refill issued"]
print("per tile of a wave (k = 0..3): median us after the first wave start")
for k in range(4):
    row = []
    for j, nm in enumerate(it_names):
        x = t[:, :, 12 + 8 * k + j]
        x = x[x > 0] - t0
        row.append(f"{nm} {np.median(x):6.2f}" if x.size else f"{nm}    -- ")
    print(f"  k={k}: " + " | ".join(row))
