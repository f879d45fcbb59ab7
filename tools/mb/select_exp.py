"""Select kernel timing (developer tool): the C2 prune-event shape (1024 windows of 2081 scores) and the local-V shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatten_amd import ops
dev = torch.device("cuda:0")
def _time(fn, n=10, reps=5):
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(n): fn(i)
        for _ in range(2): g.replay()
        side.synchronize(); t = time.perf_counter()
        for _ in range(reps): g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6
torch.manual_seed(0)
out = []
for rows, L, lo, hi, k, dt in ((1024, 4096, 4, 2085, 1020, torch.bfloat16), (1024, 4096, 4, 2085, 1020, torch.float32),
                               (40, 16384, 0, 16384, 4915, torch.bfloat16)):
    s = torch.randn(rows, L, device=dev).to(dt)
    out.append(f"{rows}x{hi - lo} {str(dt)[6:]} k={k}: {_time(lambda i: ops.topk_select(s, lo, hi, k)):.1f}us")
print("  ".join(out))
