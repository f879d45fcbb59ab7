"""Phase stamps of the one-launch local-V step (a -DSPATTEN_LV_TRACE build: bash tools/mb/build_variant.sh lvtrace local_v
-DSPATTEN_LV_TRACE; copy tools/mb/ab/lib_lvtrace.so over spatten_amd/lib/libspatten_hip.so on the box).  Cycles of head 0's splits."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops, _lib
N, H, d, dev, dt = 16384, 40, 128, torch.device("cuda"), torch.bfloat16
lib = _lib.load()
buf = torch.zeros(32 * 16, dtype=torch.int64, device=dev)
lib.spatten_debug_set_lv_trace.argtypes = [ctypes.c_void_p]
assert lib.spatten_debug_set_lv_trace(buf.data_ptr()) == 0
K = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(3)]
V = [torch.randn(1, H, N, d, device=dev, dtype=dt) for _ in range(3)]
q = torch.randn(1, H, d, device=dev, dtype=dt)
cos, sin = ops.rope_table(N + 8, d, dt, dev)
out = torch.empty(1, H * d, device=dev, dtype=dt)
stash = torch.empty(1, H, N, device=dev, dtype=dt)
keep = int(0.3 * N)
names = ["start", "keys streamed", "hist 0 built", "hist 0 read", "-", "hist 1 built", "hist 1 read", "-", "threshold", "compacted", "V gathered",
         "published", "merged"]
for it in range(4):
    ops.attn_decode_local_v(q, K[it % 3], V[it % 3], N, cos, sin, N - 1, keep, stash, out=out)
    torch.cuda.synchronize()
    t = buf.cpu().view(32, 16)
    if it < 2:
        continue
    S = int((t[:, 0] > 0).sum())
    t0 = int(t[:S, 0].min())
    print(f"run {it}: {S} splits of head 0; cycles since the first split's start")
    for s_ in range(S):
        row = t[s_]
        base = int(row[0])           # (every XCD has its own counter: a split's stamps are relative to ITS start)
        print("  split %d: " % s_ + "  ".join(f"{names[i]} {int(row[i]) - base}" for i in range(1, 13) if names[i] != "-" and int(row[i]) > 0))
