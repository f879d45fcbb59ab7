"""Phase stamps of the fused projection + attention launch (needs a library built with -DSPATTEN_TRACE -DSPATTEN_TRACE_SLOTS=16:
tools/mb/fused_trace.sh).  Per workgroup, cycles since the EARLIEST workgroup start of the launch:
  attention team (wave 0):  0 start | 5 tile issued (after B1) | 6 q rotated (after B2) | 7 first scores | 1 tile consumed |
                            2 workgroup reduced | 4 merged (last split only)
  projection team (wave 4): 8 start | 9 weights consumed + published | 10 head's q|k|v gathered
  OPROJ=1 (the output projection inside the launch): 11 B3 passed | 12 every merged head gathered | 13 output rows stored"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from spatten_amd import _lib, ops
H, d, N = 32, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 2081
dev, dt = torch.device("cuda:0"), torch.bfloat16
hid, cap, L = H * d, 2304, 6
torch.manual_seed(0)
K = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
KR = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
V = [torch.randn(1, H, cap, d, device=dev, dtype=dt) for _ in range(L)]
W = [torch.randn(3 * hid, hid, device=dev, dtype=dt) * hid ** -0.5 for _ in range(L)]
OPROJ = os.environ.get("OPROJ", "0") == "1"
WO = [torch.randn(hid, hid, device=dev, dtype=dt) * hid ** -0.5 for _ in range(L)] if OPROJ else None
cos, sin = ops.rope_table(cap + 8, d, dt, dev)
x = torch.randn(1, 1, hid, device=dev, dtype=dt)
ws = ops.DecodeWorkspace(1, H, d, dev)
st = torch.zeros(1, H, cap, device=dev, dtype=dt)
lib = _lib.load()
lib.spatten_debug_set_trace.argtypes = [ctypes.c_void_p]
SL, WG = 16, 8 * H
tr = torch.zeros(L, WG, SL, dtype=torch.int64, device=dev)
for rnd in range(2):
    for i in range(L):
        lib.spatten_debug_set_trace(tr[i].data_ptr() if rnd else None)
        ops.attn_decode_qkv(x, W[i], None, H, K[i], KR[i], V[i], N, cos, sin, N - 1, scores=st, workspace=ws, layout=cap,
                            proj=(WO[i], None) if OPROJ else None)
torch.cuda.synchronize()
lib.spatten_debug_set_trace(None)
t = tr.cpu().numpy().astype(np.float64)
for i in range(1, L):
    a = t[i]
    rel = a - a[:, 0:1]          # cycles since the workgroup's OWN start (clocks of different XCDs are not comparable)
    def q(slot, sel=None):
        v = rel[:, slot] if sel is None else rel[sel, slot]
        v = v[a[:, slot][sel if sel is not None else slice(None)] > 0]
        return "      -" if v.size == 0 else f"{np.median(v):7.0f}/{v.max():7.0f}"
    last = np.arange(WG) % 8 == 7
    print(f"layer {i}: start {q(0)} | tile issued {q(5)} | q ready {q(6)} | scores {q(7)} | tile done {q(1)} | reduced {q(2)} | merged {q(4, last)}"
          f" || proj start {q(8)} | weights done {q(9)} | gathered {q(10)}"
          + (f" | B3 {q(11)} | heads gathered {q(12)} | y stored {q(13)}" if OPROJ else "") + "   (median/max cycles)")
