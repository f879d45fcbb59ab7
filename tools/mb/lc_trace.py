"""Phase stamps of the layer-cascade chain (head 0), per layer: 0 layer start | 1 keys built (membership + ordered keys) |
2 threshold found (radix passes) | 3 window compacted | 4 id rows written.  Needs a -DSPATTEN_LC_TRACE build of layer_cascade.hip
(tools/mb/ab/lib_LCTRACE.so); SPATTEN_LC_LEGS=1 for an undisturbed chain, 4 for the chain under the side-stream gather."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from spatten_amd import _lib, ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
L, H, d, CTX, START, RECENT, IMP = 32, 32, 128, 4096, 4, 1024, 1020
cap = 2176
K = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, CTX, d, device=dev, dtype=dt) for _ in range(L)]
sc = [torch.randn(H, CTX, device=dev, dtype=dt) for _ in range(L)]
cos, sin = ops.rope_table(CTX, d, dt, dev)
hi = CTX - RECENT
keeps = [IMP - (IMP // 2) * l // (L - 1) for l in range(L)]
lib = _lib.load()
lib.spatten_debug_set_lc_trace.argtypes = [ctypes.c_void_p]
tr = torch.zeros(L, 8, dtype=torch.int64, device=dev)
run = lambda: ops.prune_layer_cascade(sc, [None] * L, 0, K, V, [CTX] * L, [hi] * L, keeps, START, [cap] * L, (cos, sin))
run(); torch.cuda.synchronize()
lib.spatten_debug_set_lc_trace(tr.data_ptr())
run(); torch.cuda.synchronize()
lib.spatten_debug_set_lc_trace(None)
t = tr.cpu().numpy().astype(np.float64)
dl = np.diff(t[:, :5], axis=1)
print("cycles per phase (keys, select, compact, ids), median over layers:", np.median(dl, axis=0).round(0), " per layer:", np.median(t[1:, 0] - t[:-1, 0]).round(0))
print("layers 1, 8, 16, 31:", dl[[1, 8, 16, 31]].round(0).tolist())
