"""Where a turn's fixed cost goes in DecodeGraph (reserve / warm-up step / capture / first replay), and the single-token
projection kernel alone at the two Llama-2-7B shapes (HIP-graph timing over rotating weight copies)."""
import contextlib
import io
import os
import sys
import time
from types import SimpleNamespace

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spatten_amd import enable_spatten_llm, ops  # noqa: E402
from spatten_amd.graph import DecodeGraph  # noqa: E402

dev, dt = torch.device("cuda", 0), torch.bfloat16
HEADS, HEAD_DIM, LAYERS = bench.HEADS, bench.HEAD_DIM, bench.LAYERS
hid = HEADS * HEAD_DIM


def t_us(fn, n=20, reps=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(0)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(n):
                fn(i)
        g.replay()
        side.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            g.replay()
        side.synchronize()
    return (time.perf_counter() - t) / (n * reps) * 1e6


with torch.no_grad():
    x = torch.randn(1, 1, hid, device=dev).to(dt)
    for N in (hid, 3 * hid):
        Ws = [(torch.randn(N, hid, device=dev) * hid ** -0.5).to(dt) for _ in range(8)]       # 8 x 33 / 100 MB > the MALL
        y = torch.empty(1, 1, N, dtype=dt, device=dev)
        a = t_us(lambda i: ops.gemv(x, Ws[i % 8], out=y))
        b = t_us(lambda i: torch.nn.functional.linear(x, Ws[i % 8]))
        mb = N * hid * 2 / 1e6
        print(f"gemv N={N}: {a:.2f} us = {mb / a * 1e3:.0f} GB/s ; torch linear {b:.2f} us = {mb / b * 1e3:.0f} GB/s")
        del Ws

    class LlamaAttention(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(pretraining_tp=1)
            self.num_heads = self.num_key_value_heads = HEADS
            self.num_key_value_groups, self.head_dim, self.hidden_size = 1, HEAD_DIM, hid
            for nme in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(self, nme, nn.Linear(hid, hid, bias=False, dtype=dt, device=dev))

    class Stack(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(model_type="llama")
            self.layers = nn.ModuleList([LlamaAttention() for _ in range(LAYERS)])
    model = Stack()
    with contextlib.redirect_stdout(io.StringIO()):
        enable_spatten_llm(model, 4, 1020, 1024, prefill_stash=False, assume_causal=True, fuse_qkv=True, native_gemv=True)
    P = 2048
    xp = torch.randn(1, P, hid, device=dev).to(dt)
    mask = torch.zeros(1, 1, P, P, dtype=dt, device=dev).masked_fill_(torch.ones(P, P, dtype=torch.bool, device=dev).triu(1), torch.finfo(dt).min)
    pos = torch.arange(P, device=dev)[None]
    past = [m(xp, attention_mask=mask, position_ids=pos, past_key_value=None, use_cache=True)[2] for m in model.layers]

    def step_fn(pst, xin):
        n = pst[0][0].shape[2]
        zm = torch.zeros(1, 1, 1, n + 1, dtype=dt, device=dev)
        pid = torch.full((1, 1), n, dtype=torch.long, device=dev)
        new, o = [], None
        for i, m in enumerate(model.layers):
            o, _, kv = m(xin, attention_mask=zm, position_ids=pid, past_key_value=pst[i], use_cache=True)
            new.append(kv)
        return new, o
    for rep in range(3):
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        g = DecodeGraph(step_fn, past, horizon=64)
        torch.cuda.synchronize(); ts.append(time.perf_counter())
        for _ in range(4):
            g.step(x)
            torch.cuda.synchronize(); ts.append(time.perf_counter())
        past = g.past_key_values
        torch.cuda.synchronize(); ts.append(time.perf_counter())
        print("ms: init %.2f | step1 (eager) %.2f | step2 (capture+replay) %.2f | step3 %.2f | step4 %.2f | views %.2f" %
              tuple((b - a) * 1e3 for a, b in zip(ts, ts[1:])))
