"""Quick on-GPU timing probe for the decode and prune kernels (developer tool, not part of the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spatten_amd import ops

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

def main():
    dt = torch.bfloat16
    B, H, d, L = 1, 32, 128, 32
    for N in (2048, 4096):
        caches = [(torch.randn(B, H, N + 64, d, device="cuda", dtype=dt), torch.randn(B, H, N + 64, d, device="cuda", dtype=dt)) for _ in range(L)]
        q = torch.randn(B, H, d, device="cuda", dtype=dt)
        cos, sin = ops.rope_table(N + 64, d, dt, "cuda")
        scores = torch.empty(B, H, N + 64, device="cuda", dtype=dt)
        out = torch.empty(B, H * d, device="cuda", dtype=dt)
        for ns in (0, 4, 8, 16, 32):
            def step():
                for kc, vc in caches:
                    ops.attn_decode(q, kc, vc, N, cos, sin, N - 1, out=out, scores=scores, n_splits=ns)
            g = torch.cuda.CUDAGraph()
            step(); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                step()
            t = timeit(g.replay, iters=20) / L
            byts = 2 * B * H * N * d * 2
            print(f"decode N={N} splits={ns}: {t:.2f} us/layer (graph, incl. gaps)  {byts / t / 1e6:.2f} TB/s")
        del caches
    # prune: 32 layers batched
    N = 4096
    Ks = [torch.randn(B, H, N, d, device="cuda", dtype=dt) for _ in range(L)]
    Vs = [torch.randn(B, H, N, d, device="cuda", dtype=dt) for _ in range(L)]
    sc = [torch.randn(H, N, device="cuda", dtype=dt) for _ in range(L)]
    Kd = [torch.empty(B, H, 2112, d, device="cuda", dtype=dt) for _ in range(L)]
    Vd = [torch.empty(B, H, 2112, d, device="cuda", dtype=dt) for _ in range(L)]
    plan = ops.PrunePlan(sc, Ks, Vs, Kd, Vd)
    idx = torch.empty(L, H, 1020, dtype=torch.int32, device="cuda")
    def prune():
        ops.prune_layers(sc, Ks, Vs, N, 4, 3072, 1020, dst=(Kd, Vd), plan=plan, idx=idx)
    t = timeit(prune, iters=10)
    byts = 4 * L * B * H * 2048 * d * 2
    print(f"prune 32 layers (select+compact): {t:.1f} us  -> {byts / t / 1e6:.2f} TB/s on compact bytes")
    lib = ops._lib.load()
    # select only / compact only
    def sel():
        for l in range(L):
            ops.topk_select(sc[l], 4, 3072, 1020)
    print(f"select per-layer launches x32: {timeit(sel, iters=5):.1f} us")

if __name__ == "__main__":
    main()
