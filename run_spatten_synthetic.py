#!/usr/bin/env python3
"""Multi-turn driver for the MI355X path — counterpart of the reference's ``run_spatten_llama.py`` for a box
without model weights or network: a random-weight stack of Llama-geometry attention layers stands in for the
checkpoint, random token ids for the MT-Bench prompts.  The caller protocol is the reference's
(run_spatten_llama.py:60-87): ``enable_spatten_llm`` -> for every turn {prune at the turn boundary from the last
step's stashed scores, prefill the prompt, greedy-decode ``max_gen_len`` tokens}, printing the reference's
"N pruned token this round" counters.

    python run_spatten_synthetic.py --layers 8 --turns 5
    python run_spatten_synthetic.py --layers 32 --turns 4 --auto-graph                            # the same loop, one graph replay per token
    python run_spatten_synthetic.py --layers 8 --cascade --head-keep 24 --pq-threshold 0.05     # SpAtten modes, through the plugin
    python run_spatten_synthetic.py --layers 12 --schedule tests/golden/trace_synthetic.csv      # per-layer keeps from a trace
    python run_spatten_synthetic.py --prompts tests/golden/mt_bench_sample.jsonl                 # MT-Bench turn structure
    python run_spatten_synthetic.py --trace tests/golden/trace_synthetic.csv --kv-len 4096     # cascade schedule demo, one K/V pair

``--trace`` reads a schedule in the format of the reference's ``spatten_hardware/hardware/workloads/*.csv``
(spatten_amd/traces.py) and applies its per-layer token / local-V / head keep ratios and requant threshold to one
decode step per layer through the cascade entry points (parity-unpinned semantics).
"""
import argparse
import time
from types import SimpleNamespace

import torch
from torch import nn


class LlamaAttention(nn.Module):                      # duck-typed by class name, like HF's module
    def __init__(self, heads, head_dim, dtype):
        super().__init__()
        hid = heads * head_dim
        self.config = SimpleNamespace(pretraining_tp=1)
        self.num_heads = self.num_key_value_heads = heads
        self.num_key_value_groups = 1
        self.head_dim, self.hidden_size = head_dim, hid
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin = nn.Linear(hid, hid, bias=False, dtype=dtype)
            nn.init.normal_(lin.weight, std=hid ** -0.5)
            setattr(self, n, lin)


class Block(nn.Module):
    def __init__(self, heads, head_dim, dtype):
        super().__init__()
        self.self_attn = LlamaAttention(heads, head_dim, dtype)


class SyntheticLlama(nn.Module):
    def __init__(self, layers, heads, head_dim, vocab, dtype):
        super().__init__()
        hid = heads * head_dim
        self.config = SimpleNamespace(model_type="llama")
        self.embed = nn.Embedding(vocab, hid, dtype=dtype)
        self.layers = nn.ModuleList([Block(heads, head_dim, dtype) for _ in range(layers)])
        self.lm_head = nn.Linear(hid, vocab, bias=False, dtype=dtype)
        self.dtype = dtype

    @torch.no_grad()
    def forward(self, input_ids=None, past_key_values=None, use_cache=True):
        B, q = input_ids.shape
        P = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        pos = torch.arange(P, P + q, device=input_ids.device)[None]
        mask = None
        if q > 1:      # HF 4.33 additive causal mask [B,1,q,N]
            i = torch.arange(q, device=input_ids.device)[:, None]
            j = torch.arange(P + q, device=input_ids.device)[None, :]
            mask = torch.where(j <= P + i, 0.0, torch.finfo(self.dtype).min).to(self.dtype)[None, None].expand(B, 1, q, P + q)
        x = self.embed(input_ids)
        new_past = []
        for i, blk in enumerate(self.layers):
            a, _, kv = blk.self_attn(x, attention_mask=mask, position_ids=pos,
                                     past_key_value=None if past_key_values is None else past_key_values[i], use_cache=True)
            x = (x + a) * 0.7071
            new_past.append(kv)
        return SimpleNamespace(logits=self.lm_head(x[:, -1:]), past_key_values=new_past)


@torch.no_grad()
def greedy_generate(model, input_ids, past_key_values, max_gen_len):          # run_spatten_llama.py:18-57
    out = model(input_ids=input_ids, past_key_values=past_key_values, use_cache=True)
    past_key_values = out.past_key_values
    tok = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
    generated = [tok.item()]                                                  # the reference reads every token on the host
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max_gen_len - 1):
        out = model(input_ids=tok, past_key_values=past_key_values, use_cache=True)
        past_key_values = out.past_key_values
        tok = out.logits[:, -1, :].argmax(dim=-1).unsqueeze(1)
        generated.append(tok.item())
    return past_key_values, len(generated), (max_gen_len - 1) / max(time.perf_counter() - t0, 1e-9)


def chat(args):
    from spatten_amd import enable_spatten_llm
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    torch.manual_seed(0)
    model = SyntheticLlama(args.layers, args.heads, args.head_dim, args.vocab, dt).cuda()
    ext = {}
    if args.cascade:
        ext["importance_mode"] = "cascade"
    if args.head_keep:
        ext["head_keep"] = args.head_keep
    if args.pq_threshold is not None:
        ext["pq_threshold"] = args.pq_threshold
    if args.local_v_keep is not None:
        ext["local_v_keep"] = args.local_v_keep
    if args.schedule:
        # per-layer schedule in the reference's workloads/*.csv format (spatten_amd/traces.py): the token keep ratios
        # become layer_keep (layer-to-layer cascade), the head ratios head_keep, the requant threshold pq_threshold
        from spatten_amd.traces import read_trace
        sched = read_trace(args.schedule)
        fr = sched.fractions(0)
        if sched.token_scope(0) == "global":      # one key_fetch_num per layer for all heads: ONE kept set per layer
            ext["token_scope"] = "global"
        fr = [fr[min(i * len(fr) // args.layers, len(fr) - 1)] for i in range(args.layers)]      # stretch to our depth
        top = max(f["token_keep"] for f in fr)
        keeps, heads = [], []
        for f in fr:
            keeps.append(max(8, min(args.important_size, int(round(args.important_size * f["token_keep"] / top)))))
            heads.append(max(1, int(round(args.heads * f["head_keep"]))))
        ext["layer_keep"] = [min(keeps[:i + 1]) for i in range(len(keeps))]
        if min(heads) < args.heads:
            ext["head_keep"] = [min(heads[:i + 1]) for i in range(len(heads))]
        thr = [f["requant_threshold"] for f in fr if f["requant_threshold"] is not None]
        if thr and args.pq_threshold is None and args.local_v_keep is None:
            ext["pq_threshold"] = float(thr[0])
            prof = read_trace(args.schedule).pq_profile(0)         # the trace's bit columns: key MSB bits / value bits
            if prof in ((4, 8), (8, 8), (6, 6)) and not args.cascade:
                ext["pq_profile"] = prof
        print("schedule:", {k: v for k, v in ext.items()})
    if args.auto_graph:       # the loop below stays the reference's: its single-token calls replay one captured graph per token
        ext.update(auto_graph=True, fuse_qkv=True, native_gemv=True)
    kv_cache = enable_spatten_llm(model, args.start_size, args.important_size, args.recent_size, **ext)     # :110-115
    attn = [m for m in model.modules() if type(m).__name__ == "LlamaAttention"]
    gen = torch.Generator(device="cuda").manual_seed(1)
    prompts = None
    if args.prompts:                       # MT-Bench question.jsonl: the turn structure of the reference's demo (:104-107);
        from spatten_amd.utils import load_mt_bench_prompts       # no tokenizer here: one id per whitespace token
        prompts = [[hash(w) % args.vocab for w in t.split()] or [0] for t in load_mt_bench_prompts(args.prompts)]
    past, cumulative = None, 0
    for idx in range(len(prompts) if prompts else args.turns):
        if prompts:
            input_ids = torch.tensor([prompts[idx]], device="cuda")
            plen = input_ids.shape[1]
        else:
            plen = int(torch.randint(args.prompt_len // 2, args.prompt_len + 1, (1,)).item())
            input_ids = torch.randint(0, args.vocab, (1, plen), device="cuda", generator=gen)
        print(f"\nUSER: <{plen} synthetic tokens>\n\nASSISTANT: ", end="")
        if past is not None:                                                                          # :71-83
            space_needed = plen + args.max_gen_len
            scores = [m.attn_scores for m in attn]
            n_prev = past[0][0].size(2)
            past = kv_cache.apply_token_pruning(past, space_needed, scores)
            pruned = n_prev - past[0][0].size(2)
            cumulative += pruned
            print(f"N pruned token this round: {pruned}, cumulative: {cumulative}")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        past, n, dec_rate = greedy_generate(model, input_ids, past, args.max_gen_len)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        lens = sorted({kv[0].size(2) for kv in past})
        print(f"<{n} tokens> kv_len={lens[0] if len(lens) == 1 else lens}  ({(plen + n) / dtm:.0f} tok/s incl. prefill, decode {dec_rate:.0f} tok/s, {args.layers} layers)")
    if kv_cache.ext is not None:
        print("extensions:", kv_cache.ext.stats())


def cascade(args):
    """One decode step per layer of a trace-driven cascade: token prune -> head prune -> local V prune -> optional
    progressive-quant keys.  Prints the per-layer configuration and device time."""
    from spatten_amd import kv_slab, ops
    from spatten_amd.cascade import CascadeImportance, HeadPruner, local_v_decode
    from spatten_amd.traces import read_trace
    sched = read_trace(args.trace)
    fr = sched.fractions(0)
    prof = sched.pq_profile(0)            # (key MSB bits, value bits) of the trace, TestSpAtten.scala:64-97
    dt, H, d, N = torch.bfloat16, args.heads, args.head_dim, args.kv_len
    gen = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen).to(dt)
    cos, sin = kv_slab.rope_tables(N + 64, d, dt, "cuda")
    K, V = rnd(1, H, N, d), rnd(1, H, N, d)
    Kr = ops.rope_single(K, cos, sin)
    ci = CascadeImportance(1, H, N, "cuda")
    hp = HeadPruner(H, "cuda")
    q = rnd(1, H, d)
    stash = torch.empty(1, H, N, dtype=dt, device="cuda")
    lse = torch.empty(1, H, 2, dtype=torch.float32, device="cuda")
    n = N
    print(f"{'layer':>5} {'keys':>6} {'values':>6} {'heads':>5} {'pq_thr':>6} {'us':>8}")
    for f in fr:
        keep = max(int(round(f["token_keep"] * N)), 16)
        if keep < n:                                  # global token pruning on the ACCUMULATED importance
            idx = ci.select(0, n, 4, n, keep - 4)
            K, V, Kr = ops.kv_compact(K, V, idx, 4, n, L=n, rope=(cos, sin))
            ci.compact(0, idx, 4, n, n)
            n = keep
            stash = torch.empty(1, H, n, dtype=dt, device="cuda")
        heads = hp.select(max(int(round(f["head_keep"] * H)), 1)) if f["head_keep"] < 1.0 else None
        vkeep = max(int(round(f["value_keep"] * n)), 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if f["requant_threshold"] is not None and args.pq and prof in ops.PQ_PROFILES:
            # the trace's bit profile: quantised value plane, LSB-only refetch (ops.PQProfilePlanes)
            planes = ops.PQProfilePlanes(1, H, H, n, d, "cuda", key_bits=prof[0], value_bits=prof[1])
            ops.pq_pack_planes(Kr, V, planes, 0, n)
            out = ops.attn_decode_pqv(q, planes, n, cos, sin, n - 1, f["requant_threshold"])
            ops.attn_decode(q, None, Kr, V, n, cos, sin, n - 1, scores=stash, lse=lse, scores_only=True)
        elif f["requant_threshold"] is not None and args.pq:
            planes = ops.PQPlanes(1, H, n, d, "cuda")
            ops.pq_pack(Kr, planes, 0, n)
            out = ops.attn_decode_pq(q, planes, V, n, cos, sin, n - 1, f["requant_threshold"])
            ops.attn_decode(q, None, Kr, V, n, cos, sin, n - 1, scores=stash, lse=lse, scores_only=True)
        elif vkeep < n:
            out, stash = local_v_decode(q, Kr, V, n, cos, sin, n - 1, vkeep)
            lse = ops.row_lse(stash[:, :, None, :])[:, :, 0]
        else:
            out = ops.attn_decode(q, None, Kr, V, n, cos, sin, n - 1, scores=stash, lse=lse, head_ids=heads)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6
        ci.accumulate(0, stash[:, :, None, :], lse[:, :, None, :])
        hp.observe(torch.nan_to_num(out)[:, None, :])
        q = rnd(1, H, d)
        thr = f["requant_threshold"]
        print(f"{f['layer']:>5} {n:>6} {vkeep:>6} {H if heads is None else heads.numel():>5} {('-' if thr is None else thr):>6} {us:>8.1f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--turns", type=int, default=5)
    ap.add_argument("--prompt-len", type=int, default=160)
    ap.add_argument("--max-gen-len", type=int, default=64)
    ap.add_argument("--start_size", type=int, default=0)            # the reference's demo defaults, :134-136
    ap.add_argument("--important_size", type=int, default=150)
    ap.add_argument("--recent_size", type=int, default=150)
    ap.add_argument("--cascade", action="store_true", help="importance_mode='cascade' (cumulative softmax probabilities)")
    ap.add_argument("--head-keep", type=int, default=0, help="cascade head pruning: heads kept per layer")
    ap.add_argument("--pq-threshold", type=float, default=None, help="progressive quantisation: LSB refetch below this max prob")
    ap.add_argument("--local-v-keep", type=float, default=None, help="local V pruning: fraction of V rows fetched at decode")
    ap.add_argument("--auto-graph", action="store_true", help="enable_spatten_llm(auto_graph=True, fuse_qkv=True, native_gemv=True): "
                    "the unchanged per-token loop replays one captured HIP graph per token")
    ap.add_argument("--schedule", default=None, help="per-layer keeps (layer cascade / heads / requant) from a trace CSV")
    ap.add_argument("--prompts", default=None, help="MT-Bench style question.jsonl: its turns drive the chat loop")
    ap.add_argument("--trace", default=None, help="cascade schedule CSV (format of the reference's workloads/*.csv)")
    ap.add_argument("--kv-len", type=int, default=4096)
    ap.add_argument("--pq", action="store_true", help="--trace: use progressive-quant keys where the trace requants")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("run_spatten_synthetic.py needs the MI355X (there is no CPU path in the product)")
    cascade(args) if args.trace else chat(args)


if __name__ == "__main__":
    main()
