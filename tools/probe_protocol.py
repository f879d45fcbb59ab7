"""Where a turn of the reference's caller protocol goes through the plugin (run_spatten_llama.py:60-87): prune event,
prompt prefill through the patched forward, DecodeGraph bind / warm-up / capture, replays.  Llama-2-7B geometry."""
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
from spatten_amd import enable_spatten_llm  # noqa: E402
from spatten_amd.graph import DecodeGraph  # noqa: E402

dev, dt = torch.device("cuda", 0), torch.bfloat16
HEADS, HEAD_DIM, LAYERS, TURN = bench.HEADS, bench.HEAD_DIM, bench.LAYERS, bench.TURN
hid = HEADS * HEAD_DIM


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


with torch.no_grad():
    model = Stack()
    with contextlib.redirect_stdout(io.StringIO()):
        cache = enable_spatten_llm(model, 4, 1020, 1024, prefill_stash=False, assume_causal=True, fuse_qkv=True, native_gemv=True)
    P = 2048
    xp0 = torch.randn(1, P, hid, device=dev).to(dt)
    m0 = torch.zeros(1, 1, P, P, dtype=dt, device=dev)
    past = [m(xp0, attention_mask=m0, position_ids=torch.arange(P, device=dev)[None], past_key_value=None, use_cache=True)[2] for m in model.layers]
    xt = torch.randn(1, 1, hid, device=dev).to(dt)
    xp = torch.randn(1, TURN, hid, device=dev).to(dt)

    def step_fn(pst, xin):
        n = pst[0][0].shape[2]
        zm = torch.zeros(1, 1, 1, n + 1, dtype=dt, device=dev)
        pid = torch.full((1, 1), n, dtype=torch.long, device=dev)
        new, o = [], None
        for i, m in enumerate(model.layers):
            o, _, kv = m(xin, attention_mask=zm, position_ids=pid, past_key_value=pst[i], use_cache=True)
            new.append(kv)
        return new, o
    graph = DecodeGraph(step_fn, past, horizon=TURN)
    for t in range(TURN):
        graph.step(xt)
    past = graph.past_key_values
    sync = torch.cuda.synchronize
    for turn in range(3):
        sync(); t = [time.perf_counter()]
        past = cache.apply_token_pruning(past, 2 * TURN, [m.attn_scores for m in model.layers])
        sync(); t.append(time.perf_counter())
        n0 = past[0][0].shape[2]
        pm = torch.zeros(1, 1, TURN, n0 + TURN, dtype=dt, device=dev)
        pp = torch.arange(n0, n0 + TURN, device=dev)[None]
        past = [m(xp, attention_mask=pm, position_ids=pp, past_key_value=past[i], use_cache=True)[2] for i, m in enumerate(model.layers)]
        sync(); t.append(time.perf_counter())
        graph = DecodeGraph(step_fn, past, horizon=TURN)
        sync(); t.append(time.perf_counter())
        graph.step(xt); sync(); t.append(time.perf_counter())
        graph.step(xt); sync(); t.append(time.perf_counter())
        for _ in range(TURN - 3):
            graph.step(xt)
        sync(); t.append(time.perf_counter())
        past = graph.past_key_values
        sync(); t.append(time.perf_counter())
        names = ["prune", "prefill 64 (32 layers, eager)", "graph bind", "warm-up step", "capture + replay", f"{TURN - 3} replays", "views"]
        print("turn %d: " % turn + " | ".join(f"{n} {(b - a) * 1e3:.2f} ms" for n, a, b in zip(names, t, t[1:])) + f" | total {(t[-1] - t[0]) * 1e3:.1f} ms")
