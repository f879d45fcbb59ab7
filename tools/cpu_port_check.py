#!/usr/bin/env python3
"""SURVEY §8d acceptance check of the CPU baseline, run in the BUILD container (the only place the reference exists):
time the imported reference (/root/reference/spatten_llm), the torch-CPU mirror (oracle/torch_mirror.py) and the C
port (oracle/oracle.c) on the same machine, same shapes, same thread counts, and record the ratios.

    python tools/cpu_port_check.py            ->  profiles/r02_cpu_port_vs_reference.json
"""
import json
import os
import statistics
import subprocess
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the reference's `spatten_llm` is a namespace package: the repo's drop-in shim of the same name (a regular package)
# would shadow it, so the reference is imported FIRST, with the repo root off the path
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
sys.path.insert(0, "/root/reference")

import spatten_llm.kv_cache_token_pruning as ref_prune       # noqa: E402  (the reference, by import)
import spatten_llm.pos_shift.modify_llama as ref_llama       # noqa: E402

assert ref_llama.__file__.startswith("/root/reference"), ref_llama.__file__
sys.path.insert(0, ROOT)
from oracle import c_oracle as co, torch_mirror as tm        # noqa: E402

H, D, N, DT = 32, 128, 2080, torch.bfloat16                   # C2 after the prune, mid-turn
START, IMPORTANT, RECENT, CTX = 4, 1020, 1024, 4096


def med(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts) * 1e3, min(ts) * 1e3


class Rotary(nn.Module):                                      # transformers 4.33 LlamaRotaryEmbedding, restated:
    def __init__(self):                                       # the table is built once and sliced per call
        super().__init__()
        self.cached = {}

    def forward(self, x, seq_len=None):
        key = (x.dtype,)
        if key not in self.cached or self.cached[key][0].shape[2] < seq_len:
            self.cached[key] = tm.rotary_table(max(seq_len, 4096), D, x.dtype)
        c, s = self.cached[key]
        return c[:, :, :seq_len], s[:, :, :seq_len]


def stub():
    m = nn.Module()
    m.config = SimpleNamespace(pretraining_tp=1)
    m.hidden_size, m.num_heads, m.head_dim = H * D, H, D
    m.num_key_value_heads, m.num_key_value_groups = H, 1
    m.q_proj = m.k_proj = m.v_proj = m.o_proj = nn.Identity()
    m.rotary_emb = Rotary()
    return m


def main():
    torch.manual_seed(0)
    out = {"machine": {"cpus": os.cpu_count(),
                       "model": subprocess.run("lscpu | grep 'Model name' | head -1", shell=True, capture_output=True,
                                               text=True).stdout.strip().split(":")[-1].strip()},
           "shape": {"heads": H, "head_dim": D, "kv_len": N, "dtype": "bf16"}, "threads": {}}
    past_k, past_v = torch.randn(1, H, N - 1, D).to(DT), torch.randn(1, H, N - 1, D).to(DT)
    hid = torch.randn(1, 1, H * D).to(DT)
    mask = torch.zeros(1, 1, 1, N, dtype=DT)
    pos = torch.tensor([[N - 1]])
    mod = stub()
    cos, sin = tm.rotary_table(N, D, DT)
    q4 = hid.view(1, 1, H, D).transpose(1, 2)
    rs = np.random.default_rng(0)
    mk = lambda *s: (rs.standard_normal(s).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    qh, kc, vc, cs, sn = mk(1, H, D), mk(1, H, N, D), mk(1, H, N, D), mk(N, D // 2), mk(N, D // 2)
    oh, sh = np.empty((1, H * D), np.uint16), np.empty((1, H, N), np.uint16)
    # prune inputs (one layer, 4096 -> 2048)
    K4, V4 = torch.randn(1, H, CTX, D).to(DT), torch.randn(1, H, CTX, D).to(DT)
    st4 = torch.randn(1, H, 1, CTX).to(DT)
    cache = ref_prune.SpAttenKVCache(start_size=START, recent_size=RECENT, important_size=IMPORTANT)
    score = rs.standard_normal((H, CTX)).astype(np.float32)
    kfull = mk(1, H, CTX, D)
    co.load()
    for threads in (1, os.cpu_count()):
        torch.set_num_threads(threads)
        co.set_threads(threads)
        def c_prune():
            ix = co.topk_window(score, START, CTX - RECENT, IMPORTANT, "f32")
            co.kv_compact_raw("bf16", kfull, ix, START, CTX - RECENT)
            co.kv_compact_raw("bf16", kfull, ix, START, CTX - RECENT)
        legs = {
            "ref": lambda: ref_llama.llama_pos_shift_attention_forward(
                mod, hid, attention_mask=mask, position_ids=pos, past_key_value=(past_k, past_v), use_cache=True),
            "mir": lambda: tm.decode_core(q4, q4, q4, past_k, past_v, cos, sin),
            "c": lambda: co.attn_decode_raw("bf16", qh, kc, vc, cs, sn, None, oh, sh, 1, H, H, D, N, N - 1),
            "refp": lambda: cache.apply_token_pruning([(K4, V4)], 0, [st4]),
            "mirp": lambda: tm.prune_layer(K4, V4, st4, START, RECENT, IMPORTANT, 0),
            "cp": c_prune}
        best = {k: (float("inf"), float("inf")) for k in legs}
        with torch.no_grad():
            for _ in range(5):                       # interleaved rounds: a noisy shared box must not favour one leg
                for k, fn in legs.items():
                    best[k] = min(best[k], med(fn, warm=2, reps=7))
        ref_ms, mir_ms, c_ms, refp_ms, mirp_ms, cp_ms = (best[k] for k in ("ref", "mir", "c", "refp", "mirp", "cp"))
        out["threads"][str(threads)] = {
            "decode_ms_per_layer": {"reference": ref_ms[0], "torch_mirror": mir_ms[0], "c_port": c_ms[0]},
            "decode_ratio_mirror_over_reference": mir_ms[0] / ref_ms[0],
            "decode_ratio_c_port_over_reference": c_ms[0] / ref_ms[0],
            "prune_ms_per_layer": {"reference": refp_ms[0], "torch_mirror": mirp_ms[0], "c_port": cp_ms[0]},
            "prune_ratio_mirror_over_reference": mirp_ms[0] / refp_ms[0],
            "prune_ratio_c_port_over_reference": cp_ms[0] / refp_ms[0]}
    out["acceptance"] = {"rule": "SURVEY 8d: the CPU restatement used as the baseline must time within +-20 % of the "
                                 "imported reference on the same machine",
                         "torch_mirror_within_20pct": {t: abs(v["decode_ratio_mirror_over_reference"] - 1) <= 0.2
                                                       for t, v in out["threads"].items()},
                         "c_port_within_20pct": {t: abs(v["decode_ratio_c_port_over_reference"] - 1) <= 0.2
                                                 for t, v in out["threads"].items()}}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02_cpu_port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
