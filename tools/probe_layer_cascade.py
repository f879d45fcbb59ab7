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
# pre-allocated destinations: device time by events, host time per call without a synchronisation
nl_ = [START + k_ + (CTX - hi) for k_ in keeps]
Kd = [torch.empty(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
Vd = [torch.empty_like(x) for x in Kd]
Krd = [torch.empty_like(x) for x in Kd]
accs = [torch.rand(H, CTX, device=dev) for _ in range(L)]
for with_acc in (False, True):
    f = lambda: ops.prune_layer_cascade(sc, [None] * L, 0, K, V, [CTX] * L, [hi] * L, keeps, START, [cap] * L, (cos, sin),
                                        accs if with_acc else None, dst=(Kd, Vd, Krd))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t = time.perf_counter()
    for _ in range(5):
        f()
    host_ms = (time.perf_counter() - t) / 5 * 1e3
    e1.record(); torch.cuda.synchronize()
    print(f"pre-allocated planes, accumulators {with_acc}: device {e0.elapsed_time(e1) / 5:.3f} ms per event, host {host_ms:.3f} ms per call")
plan = ops.PrunePlan(sc, K, V, Kd, Vd, Krd)
idxp = torch.empty(L, H, IMP, dtype=torch.int32, device=dev)
g = lambda: ops.prune_layers(sc, K, V, CTX, START, hi, IMP, dst=(Kd, Vd, Krd), plan=plan, idx=idxp, rope=(cos, sin))
g(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g()
e1.record(); torch.cuda.synchronize()
print(f"reference-mode event, pre-allocated + plan: device {e0.elapsed_time(e1) / 5:.3f} ms")
print(f"reference-mode event (incl. allocation) {ref:.3f} ms | layer cascade, keeps 1020->510 {lc:.3f} ms | layer cascade, keeps 1020 everywhere {lc_same:.3f} ms")
