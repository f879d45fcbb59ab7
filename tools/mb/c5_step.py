"""The c5 layer-step alone (BASELINE configs[4] geometry: H = 40 of which 30 launched, d = 128, 8192 kept rows, (8,8) planes + LSB
refetch below 0.05, the step's append + pack inside the MSB pass) for PMC passes: `bench.py --config c5` under rocprofv3 --pmc
crashes inside the profiler on this stack, this 4-layer loop does not.  Prints the algorithmic bytes of a layer-step as
bench.py computes them."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spatten_amd import kv_slab, ops  # noqa: E402
dt, dev = torch.bfloat16, "cuda"
B, H, d, L, n0, keepH = 1, 40, 128, 4, 8192, 30
cap = kv_slab.round_capacity(n0 + 64)
g = torch.Generator(device=dev).manual_seed(5)
rnd = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32, generator=g).to(dt)
cos, sin = ops.rope_table(cap + 8, d, dt, dev)
ids = torch.arange(keepH, dtype=torch.int32, device=dev)
Kd, Krd, Vd, planes, need, q, kn, vn, out, stash = [], [], [], [], [], [], [], [], [], []
for l in range(L):
    k = torch.zeros(B, H, cap, d, dtype=dt, device=dev); k[:, :, :n0] = rnd(B, H, n0, d)
    v = torch.zeros_like(k); v[:, :, :n0] = rnd(B, H, n0, d)
    kr = torch.zeros_like(k)
    ops.build_shadow(k, kr, 0, n0, cos, sin)
    pl = ops.PQProfilePlanes(B, H, H, cap, d, dev, key_bits=8, value_bits=8)
    ops.pq_pack_planes(kr, v, pl, 0, n0)
    Kd.append(k); Krd.append(kr); Vd.append(v); planes.append(pl)
    need.append(torch.zeros(B * H, dtype=torch.int32, device=dev))
    q.append(rnd(B, H, d)); kn.append(rnd(B, H, d)); vn.append(rnd(B, H, d))
    out.append(torch.zeros(B, H * d, dtype=dt, device=dev)); stash.append(torch.zeros(B, H, cap, dtype=dt, device=dev))
CONF = sys.argv[1] if len(sys.argv) > 1 else "trace"       # trace: ~7 % of the heads refetch (bench.py --pq-confidence trace); uniform: all
if CONF == "trace":
    for l in range(L):
        conf = torch.rand(B, H, device=dev, generator=g) < 0.93
        kn[l] = torch.where(conf[:, :, None], (0.9 * q[l].float()).to(dt), kn[l])
ws = ops.DecodeWorkspace(B, H, d, dev)
steps = 24
for t in range(steps):
    n = n0 + t + 1
    for l in range(L):
        ops.attn_decode_pqv(q[l], planes[l], n, cos, sin, n - 1, 0.05, out=out[l], need_lsb=need[l], scores=stash[l], head_ids=ids,
                            workspace=ws, append=(kn[l], vn[l], Kd[l], Krd[l], Vd[l]))
torch.cuda.synchronize()
n_avg = n0 + (steps + 1) / 2.0
n_ref = float(sum(int(x.sum().item()) for x in need)) / L
algo = (B * keepH * n_avg * d * (8 + 8) / 8 + 2 * B * keepH * n_avg * 4 + 2 * B * keepH * d * 2 + B * keepH * n_avg * 2 + n_ref * n_avg * d / 2)
print("C5_STEP_JSON " + json.dumps({"pq_confidence": CONF, "refetch_fraction": n_ref / keepH, "layer_steps": steps * L, "heads_launched": keepH, "heads_refetched_per_step": n_ref,
                                    "avg_rows": n_avg, "algorithmic_bytes_per_layer_step": algo}))
