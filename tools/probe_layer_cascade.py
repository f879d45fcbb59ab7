"""Layer-to-layer cascade prune event at Llama-2-7B geometry (32 layers, 4096-row caches, keeps 1020 -> 510), its pieces
against the reference-mode event:  select chain + ragged gathers (ops.prune_layer_cascade)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatten_amd import ops  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
L, H, d, CTX, START, RECENT, IMP = 32, 32, 128, 4096, 4, 1024, 1020
cap = 2176
K = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
sc = [torch.randn(H, CTX, device=dev, dtype=dt) for _ in range(L)]
cos, sin = ops.rope_table(CTX, d, dt, dev)
hi = CTX - RECENT
keeps = [IMP - (IMP // 2) * l // (L - 1) for l in range(L)]


def t_ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ref = t_ms(lambda: ops.prune_layers(sc, K, V, CTX, START, hi, IMP, capacity=cap, rope=(cos, sin)))
lc = t_ms(lambda: ops.prune_layer_cascade(sc, [None] * L, 0, K, V, [CTX] * L, [hi] * L, keeps, START, [cap] * L, (cos, sin)))
lc_same = t_ms(lambda: ops.prune_layer_cascade(sc, [None] * L, 0, K, V, [CTX] * L, [hi] * L, [IMP] * L, START, [cap] * L, (cos, sin)))
print(f"reference-mode event (incl. allocation) {ref:.3f} ms | layer cascade, keeps 1020->510 {lc:.3f} ms | layer cascade, keeps 1020 everywhere {lc_same:.3f} ms")
