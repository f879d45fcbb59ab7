# session 3: a soak of the seeded sweeps and fuzzers on the final build (wider seeds than the suite's)
cd $GRAFT_REPO_ROOT
SPATTEN_SWEEP_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_random_sweep.py -x -q 2>&1 | tail -3
timeout 900 python tools/fuzz_graph.py 80 101 2>&1 | tail -2
timeout 900 python tools/fuzz_auto_graph.py 40 77 2>&1 | grep -v "top-k window" | tail -3
timeout 900 python tools/fuzz_e2e.py 16 5 2>&1 | grep -v "^ok" | tail -4
# the chained launch, many tokens in one process (epoch wrap of nothing; generations advance): 3000 chained tokens against per-layer launches
timeout 600 python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from spatten_amd import ops
B, H, d, L, N0, dt = 1, 32, 128, 8, 1500, torch.bfloat16
cap = 4800
g = torch.Generator(device="cuda").manual_seed(3)
rnd = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32, generator=g).to(dt)
cos, sin = ops.rope_table(cap + 8, d, dt, "cuda")
def planes():
    K = [torch.zeros(B, H, cap, d, dtype=dt, device="cuda") for _ in range(L)]
    V = [torch.zeros_like(k) for k in K]; Kr = [torch.zeros_like(k) for k in K]
    for l in range(L):
        K[l][:, :, :N0] = rnd(B, H, N0, d); V[l][:, :, :N0] = rnd(B, H, N0, d)
        ops.build_shadow(K[l], Kr[l], 0, N0, cos, sin)
    return K, Kr, V
g.manual_seed(3); Ka, Kra, Va = planes()
g.manual_seed(3); Kb, Krb, Vb = planes()
q = [rnd(B, H, d) for _ in range(L)]; kn = [rnd(B, H, d) for _ in range(L)]; vn = [rnd(B, H, d) for _ in range(L)]
oa = [torch.zeros(B, H * d, dtype=dt, device="cuda") for _ in range(L)]; ob = [torch.zeros_like(x) for x in oa]
ws = ops.DecodeWorkspace(B, H, d, "cuda")
chain = ops.DecodeChain(q, Kb, Krb, Vb, ob, k_new=kn, v_new=vn)
bad = 0
for t in range(3000):
    n = N0 + 1 + t
    for l in range(L):
        q[l].copy_(rnd(B, H, d)); kn[l].copy_(rnd(B, H, d)); vn[l].copy_(rnd(B, H, d))
        ops.attn_decode(q[l], Ka[l], Kra[l], Va[l], n, cos, sin, n - 1, k_new=kn[l], v_new=vn[l], out=oa[l], workspace=ws)
    chain(n, cos, sin, n - 1)
    if t % 250 == 249 or t < 3:
        torch.cuda.synchronize(); chain.check(); ws.check()
        same = all(torch.equal(a, b) for a, b in zip(oa, ob)) and all(torch.equal(a[:, :, :n], b[:, :, :n]) for a, b in zip(Kra, Krb))
        bad += not same
        print("token", t, "bit-identical" if same else "MISMATCH", flush=True)
print("chained soak:", "ok" if not bad else f"{bad} mismatches")
PY
