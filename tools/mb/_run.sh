cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | cut -c1-250
python - <<'PY'
# GQA decode: how much does a kv head's re-read by its query heads cost?  (Llama-3-8B-like: 32 q heads, 8 kv heads)
import torch, time, sys
sys.path.insert(0, '.')
from spatten_amd import ops
dev, dt, d = torch.device('cuda'), torch.bfloat16, 128
for H, Hkv, N in ((32, 32, 4096), (32, 8, 4096), (64, 8, 4096), (32, 8, 16384)):
    NC = 4
    K = [torch.randn(1, Hkv, N, d, device=dev, dtype=dt) for _ in range(NC)]
    V = [torch.randn(1, Hkv, N, d, device=dev, dtype=dt) for _ in range(NC)]
    q = torch.randn(1, H, d, device=dev, dtype=dt)
    cos, sin = ops.rope_table(N + 8, d, dt, dev)
    out = torch.empty(1, H * d, device=dev, dtype=dt)
    ws = ops.DecodeWorkspace(1, H, d, dev)
    fn = lambda i: ops.attn_decode(q, None, K[i % NC], V[i % NC], N, cos, sin, N - 1, out=out, workspace=ws)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        fn(0); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(20): fn(i)
        g.replay(); side.synchronize(); t = time.perf_counter()
        for _ in range(5): g.replay()
        side.synchronize()
    us = (time.perf_counter() - t) / 100 * 1e6
    kvb = 2 * Hkv * N * d * 2
    print(f"H={H} Hkv={Hkv} N={N}: {us:.2f} us per launch; unique K/V bytes {kvb/1e6:.1f} MB -> {kvb/us/1e6:.2f} TB/s of unique bytes, {kvb*(H//Hkv)/us/1e6:.2f} TB/s of requested bytes")
PY
