#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X: attention-path decode throughput of Llama-2-7B
geometry at N=4096 with 50% cascade token pruning (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One STEP = one decode token through the attention path of all 32 layers (fused RoPE / KV append / Q·K /
stash / softmax / P·V kernel per layer) over the pruned cache; every 64th step (a "turn", the reference's
max_gen_len, run_spatten_llama.py:61) starts with the prune event of all layers (per-head top-k over the
stashed scores of the 4096-token cache + fused gather/compaction + rotated-shadow rebuild) that takes the
cache from 4096 to 2048 rows.  Inputs are synthetic, resident in HBM before the timed region.

The timed region always STARTS AT A TURN BOUNDARY (slot 0 = the prune event): K timed steps contain ceil(K / 64)
prune events, whatever K the caller picks — never fewer than the workload's one-per-64-tokens.

N > 1: head-parallel (spatten_amd/parallel.py); `python bench.py --gpus N` spawns its own N ranks (torch.distributed.run
on a free local port) when it is not already running under a launcher.  --scaling weak (default since round 5): the batch
grows with N (B = N sequences), every rank owns H/N heads of every sequence — the same KV bytes per rank as the single-GPU
run, i.e. per-GPU work is fixed.  --scaling strong: ONE sequence (B = 1) split over the ranks, H/N heads each
(BASELINE.json configs[2] / [4]: 4 resp. 5 heads per GPU at N = 8) — total work fixed; at the c2 shape that launch is at its
latency floor (extras.per_rank_launch_us: 8.9 us for 4 heads against 11.0 for 32), so sharding ONE 2048-row sequence cannot
pay before any communication — kept measurable, no longer the default.  Either way each token all-gathers
the layers' [B, H/N*d] output slices over RCCL (`config.rccl_ranks` = what the library's communicator reports).

Defaults: --steps 512 --warmup 64 (8 turns timed; ~0.2 s of GPU time + ~25 s of CPU-baseline sampling and side
measurements); the driver may pass any K / W — the timed region always starts at a turn boundary.

Prints ONE JSON line (rank 0).  `value` = whole-job tokens/s.  `roofline` describes the dominant kernel
(decode attention, HBM-bound); `cpu_baseline` times the torch-CPU mirror of the reference's op sequence
(oracle/torch_mirror.py — it re-`cat`s and re-rotates the whole K cache every step, as the reference does) on the host
cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAYERS, HEADS, HEAD_DIM, CTX = 32, 32, 128, 4096
START, IMPORTANT, RECENT, TURN = 4, 1020, 1024, 64
# BASELINE.json configs[1] / [2] / [4].  --config c2 is the headline (and the default: a bare
# `python bench.py` measures BASELINE.json's metric on its config); c3 / c5 are the two configurations BASELINE.json names for
# 8 GPUs, valid at --gpus 1..8 like c2 (`config.workload` names what ran).
CONFIGS = {
    "c2": dict(layers=32, heads=32, ctx=4096, start=4, important=1020, recent=1024, head_keep=None, pq=None,
               workload="llama2-7b attention path: 4096-token KV cache -> per-head top-k prune to 2048 "
                        "(start 4 / important 1020 / recent 1024) -> decode, 64-token turns"),
    "c3": dict(layers=32, heads=32, ctx=4096, start=4, important=1020, recent=1024, head_keep=24, pq=None,
               workload="llama2-7b attention path (BASELINE.json configs[2]): 4096-token KV cache -> per-head top-k prune to 2048 "
                        "+ 25 % cascade head prune (24 of 32 heads survive, ranked by sum |O_h|; static ownership, pruned heads "
                        "are not launched) -> decode, 64-token turns"),
    "c5": dict(layers=40, heads=40, ctx=16384, start=4, important=4092, recent=4096, head_keep=30, pq=(8, 8), pq_threshold=0.05,
               workload="llama2-13b attention path (BASELINE.json configs[4]): 16384-token KV cache -> per-head top-k prune to 8192 "
                        "(start 4 / important 4092 / recent 4096) + 25 % cascade head prune (30 of 40) + progressive "
                        "quantisation (8-bit key MSB plane + 4-bit LSB refetch below max-prob 0.05, 8-bit value plane) -> "
                        "decode, 64-token turns"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # what a streaming copy reaches on this part (same guide; our gather / long decodes sit at 5.5-6.2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1.  strong (default; BASELINE.json configs[1] / [2] / [4] and SURVEY 8e: ONE sequence, rank r owns "
                         "heads [r H/N, (r+1) H/N) — 4 resp. 5 heads per GPU at N = 8): the same workload as the N = 1 line, "
                         "latency-floor-bound at the c2 shape.  weak: B = N sequences, every rank holds H/N heads of each (the "
                         "1-GPU KV bytes per rank: per-GPU work fixed) — a batch, not the named config; labelled as such")
    ap.add_argument("--launch", choices=["auto", "chained", "per-layer"], default="auto",
                    help="how a token's 32 (40) layer-steps are launched: 'chained' = ONE launch per token "
                         "(spatten_attn_decode_chain, round 6: a workgroup walks the layers, a layer's K/V tile is requested before "
                         "the previous layer's completion is waited for; bit-identical results), 'per-layer' = one launch per "
                         "layer.  auto: chained where it applies (16-bit keys, no collective between the layers), and the "
                         "per-layer number is measured beside it")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 (default) = BASELINE.json's headline; c3 = + 25 %% head prune (configs[2]); c5 = Llama-2-13B geometry, "
                         "16384 -> 8192 rows, head prune 30 of 40, progressive quantisation k8v8 (configs[4]).  All valid at --gpus 1..8")
    ap.add_argument("--pq-confidence", choices=["trace", "uniform"], default="trace",
                    help="c5 only — how confident the synthetic heads are.  trace (default): queries PEAKED on the step's newest key "
                         "for ~93 %% of the (layer, head) pairs, so that ~7 %% fall below the 0.05 max-probability threshold and refetch "
                         "the LSB plane — the rate of the reference's traces (323 of 4,608 head-requests, workloads/summary-gpt2-small-"
                         "wikitext2-per8.csv: auto_requant_thres / if_requant).  uniform: unit-variance logits over 8192 keys — EVERY head "
                         "is flagged, the worst case of the feature (the r05 line); measured beside the default under `extras`")
    ap.add_argument("--batch", type=int, default=0,
                    help="sequences per step (default: 1, or N under --scaling weak); lets ONE rank run the per-rank shape of a weak-"
                         "scaling run (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="skip the dense / eager comparison legs")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL path even with one rank (testing)")
    ap.add_argument("--exchange", choices=["per-layer", "flat"], default="per-layer",
                    help="N>1 with the library-owned communicator (--gather native): 'per-layer' (default) = one all-gather of "
                         "[B, H/N*d] after EVERY layer's attention launch, captured inside the token's graph — the dependency a "
                         "real decoder has (layer l+1's projections consume layer l's gathered output; the plugin's "
                         "head-parallel mode gathers in front of o_proj, modify_llama.py:146-163); 'flat' = ONE all-gather of the "
                         "32 layers' slices per token: a lower bound on the communication, not a schedule a model can run")
    ap.add_argument("--peer-store", action="store_true",
                    help="N>1: the exchange through the library's peer-store all-gather (hipIpc-mapped receive buffers, every "
                         "rank writes its slice straight into all peers: SURVEY 8e 'single-shot direct writes') instead of RCCL")
    ap.add_argument("--gather", choices=["native", "flat", "grouped", "per-layer"], default="native",
                    help="N>1, the exchange of a token's attention outputs: 'native' = ONE all-gather of the 32 layers' "
                         "slices on the library-owned RCCL communicator (spatten_comm_*), captured INSIDE the token's HIP "
                         "graph; the others go through torch.distributed eagerly after the graph: one flat all-gather, the "
                         "32 per-layer all-gathers as one RCCL group, or one collective per layer")
    return ap.parse_args()


def eager_pos_shift_layer(q, k_new, v_new, past_k, past_v, inv_freq):
    """torch-eager replica of the reference's dense pos-shift attention core (modify_llama.py:86-147) — the
    'dense HF attention' the >=4x target is measured against.  q [B,H,1,d]; past [B,H,P,d] un-rotated."""
    d = q.shape[-1]
    N = past_k.shape[2] + 1
    t = torch.arange(N, device=q.device, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(q.dtype), emb.sin().to(q.dtype)

    def rot(x, pos):
        c, s = cos[pos][None, None], sin[pos][None, None]
        h = x.shape[-1] // 2
        return x * c + torch.cat((-x[..., h:], x[..., :h]), dim=-1) * s

    qr = rot(q, torch.arange(N - 1, N, device=q.device))
    k = torch.cat([past_k, k_new], dim=2)
    v = torch.cat([past_v, v_new], dim=2)
    kr = rot(k, torch.arange(N, device=q.device))
    w = torch.matmul(qr, kr.transpose(2, 3)) / (d ** 0.5)
    stash = w.detach().clone()
    w = w + torch.zeros(q.shape[0], 1, 1, N, dtype=q.dtype, device=q.device)
    w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(q.shape[0], 1, -1)
    return o, stash, (k, v)


def time_region(fn, steps, dist_on):
    import torch.distributed as dist
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def _physical_cores() -> int:
    try:
        import subprocess
        out = subprocess.run(["lscpu", "-p=core,socket"], capture_output=True, text=True).stdout
        cores = {ln for ln in out.splitlines() if ln and not ln.startswith("#")}
        return max(1, len(cores))
    except Exception:
        return max(1, (os.cpu_count() or 2) // 2)


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _socket0_physical_cpus():
    """One logical CPU per physical core of socket 0 (lscpu), or None."""
    try:
        import subprocess
        out = subprocess.run(["lscpu", "-p=cpu,core,socket"], capture_output=True, text=True).stdout
        seen, cpus = set(), []
        for ln in out.splitlines():
            if not ln or ln.startswith("#"):
                continue
            cpu, core, sock = (int(x) for x in ln.split(",")[:3])
            if sock == 0 and core not in seen:
                seen.add(core)
                cpus.append(cpu)
        return cpus or None      # (not intersected with this process's affinity: a runtime may have pinned the calling thread)
    except Exception:
        return None


def cpu_baseline_child(L, new_len, lo, hi):
    """The timing half of `cpu_baseline`, run in a CHILD process that the parent pinned to the physical cores of one socket
    with OMP_PROC_BIND=close / OMP_PLACES=cores in its environment (the same mirror gave 12.0 / 2.6 / 41
    tokens/s on three driver hosts with free-floating threads).  Per thread count: three repetitions of a median-of-many; the
    value is the median of the three, the spread (max - min) / median is reported beside it."""
    import statistics

    import numpy as np
    from oracle import c_oracle as co, torch_mirror as tm

    H, d, n = HEADS, HEAD_DIM, new_len + TURN // 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    q4 = torch.randn(1, H, 1, d, generator=g).to(dt)
    pk, pv = torch.randn(1, H, n - 1, d, generator=g).to(dt), torch.randn(1, H, n - 1, d, generator=g).to(dt)
    K4, V4 = torch.randn(1, H, CTX, d, generator=g).to(dt), torch.randn(1, H, CTX, d, generator=g).to(dt)
    st4 = torch.randn(1, H, 1, CTX, generator=g).to(dt)
    cos, sin = tm.rotary_table(n, d, dt)

    def med(fn, budget_s, warm=3, min_reps=5):
        for _ in range(warm):
            fn()
        ts, t_end = [], time.perf_counter() + budget_s
        while len(ts) < min_reps or (time.perf_counter() < t_end and len(ts) < 4096):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return statistics.median(ts), len(ts)

    ncpu = len(os.sched_getaffinity(0))
    counts = sorted({ncpu, min(32, ncpu), min(8, ncpu), 1}, reverse=True)
    tps = lambda t_dec, t_pr: 1.0 / (L * t_dec + L * t_pr / TURN)
    legs = {}
    with torch.no_grad():
        for threads in counts:
            torch.set_num_threads(threads)
            reps = []
            for _ in range(3):
                t_dec, n_dec = med(lambda: tm.decode_core(q4, q4, q4, pk, pv, cos, sin), 1.0)
                t_pr, n_pr = med(lambda: tm.prune_layer(K4, V4, st4, START, RECENT, IMPORTANT, 0), 0.35, warm=1, min_reps=3)
                reps.append((tps(t_dec, t_pr), t_dec, t_pr, n_dec, n_pr))
            reps.sort()
            mid = reps[1]
            legs[threads] = {"value": round(mid[0], 4), "ms_per_layer_decode": round(mid[1] * 1e3, 3), "ms_per_layer_prune": round(mid[2] * 1e3, 3),
                             "repetitions": [round(r[0], 4) for r in reps], "spread": round((reps[-1][0] - reps[0][0]) / mid[0], 4),
                             "n_decode": sum(r[3] for r in reps), "n_prune": sum(r[4] for r in reps)}
    # The REPORTED baseline (round 6): the torch mirror — the reference's own op sequence
    # (kv_cache_token_pruning.py:42-96 + modify_llama.py:86-147) — on ALL physical cores of one socket, pinned; median of three
    # repetitions, their spread beside it.  (r05 reported the C port on one thread: the most repeatable leg, but 35x slower than
    # the socket runs the reference's ops.)  Every other leg stays on the line: by_threads, best_of_thread_counts, one_thread.
    best = max(counts, key=lambda c_: legs[c_]["value"])
    full = legs[ncpu]
    out = {"value": full["value"], "unit": "tokens/s", "cores": ncpu, "kind": "port",
           "sample": f"{full['n_decode']} decode-attention layer steps at kv_len {n} + {full['n_prune']} one-layer prune events "
                     f"({CTX} -> {new_len}) in 3 repetitions (median of the repetitions' medians; spread = (max - min) / median), torch-CPU "
                     f"mirror of the reference's op sequence (oracle/torch_mirror.py), extrapolated to {L} layers per token and one prune "
                     f"per {TURN} tokens; process pinned to the {ncpu} physical cores of one socket, {ncpu} threads, OMP_PROC_BIND=close",
           "ms_per_layer_decode": full["ms_per_layer_decode"], "ms_per_layer_prune": full["ms_per_layer_prune"],
           "spread": full["spread"], "repetitions": full["repetitions"],
           "best_of_thread_counts": {"threads": best, "value": legs[best]["value"], "spread": legs[best]["spread"]},
           "by_threads": {str(c_): legs[c_] for c_ in counts}, "one_thread": {"torch_mirror": legs[1]},
           "pinned_cpus": ncpu, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES")}
    try:      # the C port beside it (-march=native build on this host when gcc is there)
        co.load(native=True)
        rs = np.random.default_rng(0)
        mk = lambda *s: (rs.standard_normal(s).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        qh, kc, vc, cs, sn = mk(1, H, d), mk(1, H, n, d), mk(1, H, n, d), mk(n, d // 2), mk(n, d // 2)
        oh, sh = np.empty((1, H * d), np.uint16), np.empty((1, H, n), np.uint16)
        score, kfull = rs.standard_normal((H, CTX)).astype(np.float32), mk(1, H, CTX, d)

        def c_prune():
            ix = co.topk_window(score, lo, hi, IMPORTANT, "f32")
            co.kv_compact_raw("bf16", kfull, ix, START, hi)
            co.kv_compact_raw("bf16", kfull, ix, START, hi)
        cport = {}
        for threads in (min(ncpu, H), 1):
            co.set_threads(threads)
            reps = []
            for _ in range(3):
                td, nd = med(lambda: co.attn_decode_raw("bf16", qh, kc, vc, cs, sn, None, oh, sh, 1, H, H, d, n, n - 1), 0.8)
                tp, npr = med(c_prune, 0.3, warm=1, min_reps=3)
                reps.append((tps(td, tp), td, tp, nd, npr))
            reps.sort()
            mid = reps[1]
            cport[str(threads)] = {"value": round(mid[0], 4), "ms_per_layer_decode": round(mid[1] * 1e3, 3),
                                   "ms_per_layer_prune": round(mid[2] * 1e3, 3), "repetitions": [round(r[0], 4) for r in reps],
                                   "spread": round((reps[-1][0] - reps[0][0]) / mid[0], 4),
                                   "n_decode": sum(r[3] for r in reps), "n_prune": sum(r[4] for r in reps)}
        out["c_port_by_threads"] = cport
        out["one_thread"]["c_port"] = cport["1"]       # (oracle/oracle.c, -march=native: repeats to < 1 % across hosts)
    except Exception as e:
        out["c_port_error"] = f"{type(e).__name__}: {e}"
    print("CPU_BASELINE_JSON " + json.dumps(out), flush=True)




def cpu_baseline(L, new_len, lo, hi):
    """The reference path on the host CPU (checker code from oracle/, used here only as the timed baseline).  The timing runs
    in a child process pinned to one socket's physical cores (cpu_baseline_child); the kept-set comparison needs the GPU and
    runs here."""
    import subprocess
    H, d = HEADS, HEAD_DIM
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    cpus = _socket0_physical_cpus()
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    if cpus:
        env["SPATTEN_BENCH_PIN_CPUS"] = ",".join(str(c_) for c_ in cpus)      # the child pins itself before its first parallel region
    pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], capture_output=True, text=True, env=env,
                        timeout=900)
    line = [ln for ln in pr.stdout.splitlines() if ln.startswith("CPU_BASELINE_JSON ")]
    if pr.returncode != 0 or not line:
        raise RuntimeError(f"cpu baseline child failed: {pr.stderr[-800:]}")
    out = json.loads(line[-1][len("CPU_BASELINE_JSON "):])
    out.update({"cpu_model": _cpu_model(), "logical_cpus": os.cpu_count(), "physical_cores": _physical_cores(),
                "accepted_vs_reference": "profiles/r02_cpu_port_vs_reference.json (mirror / imported reference = 0.87-0.97 decode, "
                                         "0.89-0.98 prune, build container, 1 and 8 threads)"})
    from oracle import torch_mirror as tm
    try:
        # ---- how often does the KEPT SET differ from the reference's at C2 scale?  One decode step on
        # a 4095-token cache: the mirror's stash (the reference's op sequence, bf16 on the host) -> the reference's top-k
        # (kv_cache_token_pruning.py:59-63) against the HIP kernel's stash -> the HIP select, per head
        if torch.cuda.is_available():
            from spatten_amd import ops
            with torch.no_grad():
                qc, kcn, vcn = (torch.randn(1, H, 1, d, generator=g).to(dt) for _ in range(3))
                pkc = (torch.randn(1, H, CTX - 1, d, generator=g) * d ** -0.5).to(dt)
                pvc = torch.randn(1, H, CTX - 1, d, generator=g).to(dt)
                cosf, sinf = tm.rotary_table(CTX, d, dt)
                _, st_ref, _ = tm.decode_core(qc, kcn, vcn, pkc, pvc, cosf, sinf)
                sel = st_ref.sum(0).sum(1)[:, START:CTX - RECENT]
                idx_ref = torch.topk(sel, IMPORTANT, dim=-1).indices.sort().values + START
                dev_ = torch.device("cuda", torch.cuda.current_device())
                cg, sg = ops.rope_table(CTX, d, dt, dev_)
                kd = torch.zeros(1, H, CTX, d, dtype=dt, device=dev_)
                vd, krd = torch.zeros_like(kd), torch.zeros_like(kd)
                kd[:, :, :CTX - 1], vd[:, :, :CTX - 1] = pkc.to(dev_), pvc.to(dev_)
                ops.build_shadow(kd, krd, 0, CTX - 1, cg, sg)
                st_gpu = torch.empty(1, H, CTX, dtype=dt, device=dev_)
                ops.attn_decode(qc[:, :, 0].to(dev_), kd, krd, vd, CTX, cg, sg, CTX - 1, k_new=kcn[:, :, 0].to(dev_),
                                v_new=vcn[:, :, 0].to(dev_), scores=st_gpu)
                idx_gpu = ops.topk_select(st_gpu[0], START, CTX - RECENT, IMPORTANT).cpu()
                st_g = st_gpu.cpu()[:, :, None, :]
                same = [len(set(idx_ref[h].tolist()) & set(idx_gpu[h].tolist())) for h in range(H)]
                out["kept_set_vs_reference_c2"] = {
                    "heads": H, "heads_with_identical_kept_set": int(sum(x == IMPORTANT for x in same)),
                    "min_overlap": int(min(same)), "of": IMPORTANT,
                    "stash_entries_differing": round(float((st_g != st_ref).float().mean()), 6),
                    "how": "one decode step on a 4095-token cache, bf16: reference op sequence (oracle/torch_mirror.py) -> torch.topk "
                           "against the HIP stash -> spatten_topk_select; a differing stash entry (<= 2 ulp) can only swap tokens "
                           "that sit AT the threshold, where torch.topk's own choice among equal scores is unspecified"}
    except Exception as e:      # noqa: BLE001 - informative only
        out["kept_set_vs_reference_c2"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def plugin_path_tokens_per_s(dev, dt, n_tokens=48, variants=((False, False, False), (True, False, False), (True, True, False),
                                                              (True, True, True), (True, True, "fused"))):
    """The DROP-IN number: tokens/s through the patched HF forward itself (`llama_pos_shift_attention_forward`, called per
    layer with the arguments transformers 4.33 passes — hidden states, a zero mask, position_ids, the layer's (K, V)
    pair — including the module's q/k/v/o projections and every per-call host step), eager launches, Llama-2-7B geometry,
    2048-row pruned cache.  Next to it the same with enable_spatten_llm(assume_causal=True) and with fuse_qkv=True on top."""
    from types import SimpleNamespace

    from torch import nn

    from spatten_amd import enable_spatten_llm, ops

    class LlamaAttention(nn.Module):
        def __init__(self):
            super().__init__()
            hid = HEADS * HEAD_DIM
            self.config = SimpleNamespace(pretraining_tp=1)
            self.num_heads = self.num_key_value_heads = HEADS
            self.num_key_value_groups, self.head_dim, self.hidden_size = 1, HEAD_DIM, hid
            for nme in ("q_proj", "k_proj", "v_proj", "o_proj"):
                lin = nn.Linear(hid, hid, bias=False, dtype=dt, device=dev)
                nn.init.normal_(lin.weight, std=hid ** -0.5)
                setattr(self, nme, lin)

    class Stack(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(model_type="llama")
            self.layers = nn.ModuleList([LlamaAttention() for _ in range(LAYERS)])

    out = {}
    P = START + IMPORTANT + RECENT
    with torch.no_grad():
        model = Stack()
        for flag, fuse, gemv in variants:
            import contextlib
            import io
            fused_step = gemv == "fused"        # round 4: the q/k/v projections inside the attention launch (65 launches per token)
            gemv = bool(gemv)
            with contextlib.redirect_stdout(io.StringIO()):       # the constructor prints the reference's banner
                cache = enable_spatten_llm(model, START, IMPORTANT, RECENT, prefill_stash=False, assume_causal=flag, fuse_qkv=fuse,
                                   native_gemv=gemv, fused_step=fused_step)
            hid = HEADS * HEAD_DIM
            x = torch.randn(1, P, hid, device=dev, dtype=torch.float32).to(dt)
            mask = torch.zeros(1, 1, P, P, dtype=dt, device=dev).masked_fill_(
                torch.ones(P, P, dtype=torch.bool, device=dev).triu(1), torch.finfo(dt).min)
            pos = torch.arange(P, device=dev)[None]
            past = []
            for m in model.layers:                                   # prefill: fills every layer's cache
                _, _, kv = m(x, attention_mask=mask, position_ids=pos, past_key_value=None, use_cache=True)
                past.append(kv)
            del mask
            xt = torch.randn(1, 1, hid, device=dev, dtype=torch.float32).to(dt)

            def token(t):
                n = past[0][0].shape[2]
                zm = torch.zeros(1, 1, 1, n + 1, dtype=dt, device=dev)
                pid = torch.full((1, 1), n, dtype=torch.long, device=dev)
                for i, m in enumerate(model.layers):
                    _, _, past[i] = m(xt, attention_mask=zm, position_ids=pid, past_key_value=past[i], use_cache=True)
            for t in range(4):
                token(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(n_tokens):
                token(t)
            torch.cuda.synchronize()
            key = "plugin_path_assume_causal_fused_step_tokens_per_s" if fused_step else "plugin_path_assume_causal_fused_qkv_native_gemv_tokens_per_s" if gemv else (
                "plugin_path_assume_causal_fused_qkv_tokens_per_s" if fuse else (
                    "plugin_path_assume_causal_tokens_per_s" if flag else "plugin_path_eager_tokens_per_s"))
            out[key] = round(n_tokens / (time.perf_counter() - t0), 2)
            if fused_step:      # (the option selected the 256-thread attention team process-wide: back to the default)
                ops.set_decode_team(512)
            # ---- the same decode step as ONE captured HIP graph of the whole layer stack (spatten_amd/graph.py): the
            # device-resident step state (ABI 3) lets a single graph replay for every token of the turn
            if (flag and not fuse) or fused_step:
                continue                # (assume_causal is implied under capture: graph legs for the plain and the fused form;
                                        #  fused_step applies to eager calls only — a traced step runs the separate launches)
            from spatten_amd.graph import DecodeGraph

            def step_fn(pst, xin):
                n = pst[0][0].shape[2]
                zm = torch.zeros(1, 1, 1, n + 1, dtype=dt, device=dev)
                pid = torch.full((1, 1), n, dtype=torch.long, device=dev)
                new, o = [], None
                for i, m in enumerate(model.layers):
                    o, _, kv = m(xin, attention_mask=zm, position_ids=pid, past_key_value=pst[i], use_cache=True)
                    new.append(kv)
                return new, o
            graph = DecodeGraph(step_fn, past, horizon=4)        # first use in the process: one-time library set-up costs
            for t in range(3):
                graph.step(xt)
            past = graph.past_key_values
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph = DecodeGraph(step_fn, past, horizon=TURN)
            for t in range(TURN):       # one turn as the reference's caller runs it: warm-up step + capture + 62 replays
                graph.step(xt)
            torch.cuda.synchronize()
            t_turn = time.perf_counter() - t0
            n_rep = 128
            t0 = time.perf_counter()
            for t in range(n_rep):
                graph.step(xt)
            torch.cuda.synchronize()
            t_rep = time.perf_counter() - t0
            sfx = "_fused_step" if fused_step else "_fused_qkv_native_gemv" if gemv else ("_fused_qkv" if fuse else "")
            out[f"plugin_path_graph{sfx}_tokens_per_s"] = round(n_rep / t_rep, 2)
            out[f"plugin_path_graph{sfx}_turn_incl_capture_tokens_per_s"] = round(TURN / t_turn, 2)
            out[f"plugin_path_graph{sfx}_recaptures_in_timed_replays"] = graph.n_binds - 1      # (none — the slabs hold them)
            if gemv and not fused_step:
                # the reference's whole caller protocol through the plugin (run_spatten_llama.py:60-87), two timed chat turns:
                # prune event from the last decode step's stashes -> prefill of a 64-token prompt through the patched
                # forward -> 63 greedy-decode steps under ONE captured graph (re-captured per turn: the prune moves the cache)
                past = graph.past_key_values
                xp = torch.randn(1, TURN, hid, device=dev, dtype=torch.float32).to(dt)
                n_tok = 0
                for turn in range(3):       # the first turn also pays the caching allocator's first hipMallocs: not timed
                    if turn == 1:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        n_tok = 0
                    past = cache.apply_token_pruning(past, 2 * TURN, [m.attn_scores for m in model.layers])
                    n0 = past[0][0].shape[2]
                    pm = torch.zeros(1, 1, TURN, n0 + TURN, dtype=dt, device=dev)
                    pm[..., n0:].masked_fill_(torch.ones(TURN, TURN, dtype=torch.bool, device=dev).triu(1), torch.finfo(dt).min)
                    pp = torch.arange(n0, n0 + TURN, device=dev)[None]
                    past = [m(xp, attention_mask=pm, position_ids=pp, past_key_value=past[i], use_cache=True)[2]
                            for i, m in enumerate(model.layers)]
                    graph = DecodeGraph(step_fn, past, horizon=TURN)
                    for t in range(TURN - 1):
                        graph.step(xt)
                    past = graph.past_key_values
                    n_tok += TURN - 1
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                out["plugin_protocol_two_turns_decode_tokens_per_s"] = round(n_tok / dt_s, 2)
                out["plugin_protocol_ms_per_turn"] = round(dt_s / 2 * 1e3, 2)
                # the same protocol with NOTHING changed in the caller's loop (run_spatten_llama.py:18-57): keyword calls of
                # model(...), outputs.past_key_values handed back, one host read per token — spatten_amd.graph.auto_graph
                # wraps model.forward (enable_spatten_llm(auto_graph=True)) and replays the captured graph underneath
                from types import SimpleNamespace as _NS
                from spatten_amd.graph import auto_graph

                class _Stack:
                    def forward(self, input_ids=None, past_key_values=None, use_cache=None):
                        n0, ql = past_key_values[0][0].shape[2], input_ids.shape[1]
                        pm = torch.zeros(1, 1, ql, n0 + ql, dtype=dt, device=dev)
                        if ql > 1:
                            pm[..., n0:].masked_fill_(torch.ones(ql, ql, dtype=torch.bool, device=dev).triu(1), torch.finfo(dt).min)
                        pp = torch.arange(n0, n0 + ql, device=dev)[None]
                        new, o = [], None
                        for i, m in enumerate(model.layers):
                            o, _, kv = m(input_ids, attention_mask=pm, position_ids=pp, past_key_value=past_key_values[i], use_cache=True)
                            new.append(kv)
                        return _NS(logits=o, past_key_values=new)

                    __call__ = lambda self, **kw: self.forward(**kw)
                stack = auto_graph(_Stack(), horizon=TURN)
                past = graph.past_key_values
                for turn in range(3):
                    if turn == 1:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        n_tok = 0
                    past = cache.apply_token_pruning(past, 2 * TURN, [m.attn_scores for m in model.layers])
                    outp = stack(input_ids=xp, past_key_values=past, use_cache=True)
                    past = outp.past_key_values
                    for t in range(TURN - 1):
                        outp = stack(input_ids=xt, past_key_values=past, use_cache=True)
                        past = outp.past_key_values
                        outp.logits[0, 0, 0].item()                     # the reference reads every token on the host
                    n_tok += TURN - 1
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                out["plugin_protocol_unchanged_loop_auto_graph_decode_tokens_per_s"] = round(n_tok / dt_s, 2)
                out["plugin_protocol_unchanged_loop_auto_graph_ms_per_turn"] = round(dt_s / 2 * 1e3, 2)
                del stack
            del graph
            del past
    # bytes one token of this path must move at least: the four projection matrices of every layer + the kept K/V rows
    hid = HEADS * HEAD_DIM
    # the drop-in surface as it is meant to be driven: enable_spatten_llm(fuse_qkv, native_gemv) with the
    # per-token model call replaced by DecodeGraph.step — one captured HIP graph of the whole patched layer stack per turn.
    # `plugin_path_eager_tokens_per_s` is the same patched forward launched op by op from Python, as the reference's loop does.
    if "plugin_path_graph_fused_qkv_native_gemv_tokens_per_s" in out:
        out["plugin_path_tokens_per_s"] = out["plugin_path_graph_fused_qkv_native_gemv_tokens_per_s"]
        out["plugin_path_config"] = ("enable_spatten_llm(fuse_qkv=True, native_gemv=True) + spatten_amd.graph.DecodeGraph (replays of one "
                                     "captured graph per turn; *_turn_incl_capture_* includes the warm-up step and the capture; "
                                     "plugin_protocol_* the whole prune -> prefill -> decode turn)")
    out["plugin_path_min_bytes_per_token"] = int(LAYERS * (4 * hid * hid * 2 + 2 * HEADS * (P + TURN) * HEAD_DIM * 2))
    out["plugin_path_tokens_per_s_at_hbm_peak"] = round(HBM_PEAK_GBS * 1e9 / out["plugin_path_min_bytes_per_token"], 1)
    return out


def dense_with_projections_tokens_per_s(dev, dt, n_tokens=6):
    """The like-for-like dense leg of the drop-in comparison: the reference forward's op sequence (modify_llama.py:72-163:
    q/k/v projections, cat, rotation of the whole key cache, QK^T/sqrt(d), stash clone, +mask, fp32 softmax, PV, o_proj) in
    eager torch on the GPU, Llama-2-7B geometry, dense N = 4096 — eagerly launched, as the reference runs it, and the same
    op sequence captured into one HIP graph per token (what a caller could do without this package)."""
    hid = HEADS * HEAD_DIM
    g = torch.Generator(device=dev).manual_seed(7)
    W = [[(torch.randn(hid, hid, device=dev, dtype=torch.float32, generator=g) * hid ** -0.5).to(dt) for _ in range(4)]
         for _ in range(LAYERS)]
    pk = [torch.randn(1, HEADS, CTX - 1, HEAD_DIM, device=dev, dtype=torch.float32, generator=g).to(dt) for _ in range(LAYERS)]
    pv = [torch.randn(1, HEADS, CTX - 1, HEAD_DIM, device=dev, dtype=torch.float32, generator=g).to(dt) for _ in range(LAYERS)]
    x = torch.randn(1, 1, hid, device=dev, dtype=torch.float32, generator=g).to(dt)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, HEAD_DIM, 2, device=dev).float() / HEAD_DIM))
    F = torch.nn.functional

    def token():
        o = None
        for l in range(LAYERS):
            wq, wk, wv, wo = W[l]
            sp = lambda t: t.view(1, 1, HEADS, HEAD_DIM).transpose(1, 2)
            a, _, _ = eager_pos_shift_layer(sp(F.linear(x, wq)), sp(F.linear(x, wk)), sp(F.linear(x, wv)), pk[l], pv[l], inv_freq)
            o = F.linear(a, wo)
        return o
    out = {}
    with torch.no_grad():
        for _ in range(2):
            token()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_tokens):
            token()
        torch.cuda.synchronize()
        out["dense_eager_with_projections_tokens_per_s"] = round(n_tokens / (time.perf_counter() - t0), 2)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                token()
                side.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    token()
                gr.replay()
                side.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_tokens):
                    gr.replay()
                side.synchronize()
            out["dense_eager_with_projections_graphed_tokens_per_s"] = round(n_tokens / (time.perf_counter() - t0), 2)
        except Exception as e:
            out["dense_graphed_error"] = f"{type(e).__name__}: {e}"
    return out


def self_spawn(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, RCCL over a free
    local port) and hand their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    global LAYERS, HEADS, CTX, START, IMPORTANT, RECENT
    args = parse()
    if args.cpu_baseline_child:         # (spawned by cpu_baseline(): pinned, no GPU work)
        pin = os.environ.get("SPATTEN_BENCH_PIN_CPUS")
        if pin:
            try:
                os.sched_setaffinity(0, [int(x) for x in pin.split(",")])
            except OSError:
                pass                    # CPUs outside this container's set: run unpinned (pinned_cpus in the output says which)
        cpu_baseline_child(LAYERS, START + IMPORTANT + RECENT, START, CTX - RECENT)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    cfg = CONFIGS[args.config]
    LAYERS, HEADS, CTX = cfg["layers"], cfg["heads"], cfg["ctx"]
    START, IMPORTANT, RECENT = cfg["start"], cfg["important"], cfg["recent"]
    headline = args.config == "c2"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:             # --force-dist without a launcher: any free port
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from spatten_amd import kv_slab, ops
    from spatten_amd.parallel import HeadParallel

    dt = torch.bfloat16
    hp = HeadParallel(HEADS)
    B, Hl, d, L = (args.batch if args.batch > 0 else (1 if args.scaling == "strong" else world)), hp.local_heads, HEAD_DIM, LAYERS
    world_eff = hp.world
    new_len = START + IMPORTANT + RECENT                     # 2048
    cap = kv_slab.round_capacity(new_len + TURN)             # 2176
    lo, hi = START, CTX - RECENT                             # window [4, 3072), num_coming = 0
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    rnd = lambda *shape: torch.randn(*shape, device=dev, dtype=torch.float32, generator=gen).to(dt)

    cos, sin = ops.rope_table(CTX + TURN, d, dt, dev)
    # ---- the 4096-token cache of every layer (this rank's heads) and the stash of its last decode step ----
    Kp = [rnd(B, Hl, CTX, d) for _ in range(L)]
    Vp = [rnd(B, Hl, CTX, d) for _ in range(L)]
    q = [rnd(B, Hl, d) for _ in range(L)]
    kn = [rnd(B, Hl, d) for _ in range(L)]
    vn = [rnd(B, Hl, d) for _ in range(L)]
    kn_uniform = None
    if cfg["pq"] is not None:
        # trace-like confidence: a confident head's NEWEST key is 0.9 x its query — relative rotary position 0, so
        # the step's logit of that key is 0.9 |q|^2 / sqrt(d) ~ 10 whatever the position, max probability ~ 0.6 >> 0.05; the other
        # ~7 % keep a random newest key: max probability ~ 1e-3 over 8192 unit-variance logits -> flagged -> LSB refetch
        kn_uniform = [x.clone() for x in kn]
        conf = [torch.rand(B, Hl, device=dev, generator=gen) < 0.93 for _ in range(L)]
        kn_trace = [torch.where(conf[l][:, :, None], (0.9 * q[l].float()).to(dt), kn[l]) for l in range(L)]
        if args.pq_confidence == "trace":
            for l in range(L):
                kn[l].copy_(kn_trace[l])
    ws = ops.DecodeWorkspace(B, Hl, d, dev)
    stash_full = [torch.empty(B, Hl, 1, CTX, dtype=dt, device=dev) for _ in range(L)]
    Krp = []
    head_sc = []
    for l in range(L):
        kr = ops.rope_single(Kp[l], cos, sin)
        o_dense = ops.attn_decode(q[l], None, kr, Vp[l], CTX, cos, sin, CTX - 1, scores=stash_full[l].view(B, Hl, CTX), workspace=ws)
        if cfg["head_keep"]:
            head_sc.append(ops.head_scores(o_dense, Hl))                     # sum |O_h| of this rank's heads (README.md:21)
        Krp.append(kr if headline else None)                                  # (the dense comparison legs of the headline run)
        del kr
    importance = [ops.importance(s).contiguous() for s in stash_full]         # [Hl, CTX] (sum over batch, :51)
    torch.cuda.synchronize()

    # ---- cascade head pruning (c3 / c5; parity unpinned, oracle: head_prune_cascade): cumulative sum |O_h| over the layers, the
    # heads of ALL ranks ranked together (one all-gather of [L, H/N] scores), a head pruned in a layer stays pruned; ownership is
    # static — a rank launches only its surviving heads (possibly none).  Decided ONCE here, from the dense step: in a session
    # the fused `head_abs` accumulation of the decode launches feeds the same rule at every turn boundary.
    hid = [None] * L
    kept_heads = None
    if cfg["head_keep"]:
        hid = hp.surviving_local_heads(torch.stack(head_sc), cfg["head_keep"])   # [L] int32 local ids (spatten_amd/parallel.py)
        kept_heads = [int(x.numel()) for x in hid]

    # ---- pruned slabs (K, rotated shadow, V) with room for one turn ------------------------------------
    Kd = [torch.zeros(B, Hl, cap, d, dtype=dt, device=dev) for _ in range(L)]
    Krd = [torch.zeros_like(x) for x in Kd]
    Vd = [torch.zeros_like(x) for x in Kd]
    plan = ops.PrunePlan(importance, Kp, Vp, Kd, Vd, Krd)
    idx = torch.empty(L, Hl, IMPORTANT, dtype=torch.int32, device=dev)
    stash = [torch.empty(B, Hl, cap, dtype=dt, device=dev) for _ in range(L)]
    # attention outputs (and all-gather receive buffers) are double-buffered by the parity of the position in the
    # turn: the RCCL gather of token t then overlaps the attention graph of token t+1 without sharing a buffer
    outs_flat = [torch.zeros(L, B, Hl * d, dtype=dt, device=dev) for _ in range(2)]   # all layers' slices of one token (a pruned head's stays 0)
    outs2 = [[outs_flat[par][l] for l in range(L)] for par in range(2)]
    outs = outs2[0]
    staging2 = [[hp.gather_staging(B, 1, d, dt, dev) for _ in range(L)] for _ in range(2)] if dist_on else None
    staging_flat = [torch.empty(world_eff * L * B * Hl * d, dtype=dt, device=dev) for _ in range(2)] if dist_on else None

    # progressive quantisation (c5): profiled planes of the pruned rows — packed by the prune event, the step's row by the step
    pq = cfg["pq"]
    planes = need = None
    if pq is not None:
        planes = [ops.PQProfilePlanes(B, Hl, Hl, cap, d, dev, key_bits=pq[0], value_bits=pq[1]) for _ in range(L)]
        need = [torch.zeros(B * Hl, dtype=torch.int32, device=dev) for _ in range(L)]

    def prune():
        ops.prune_layers(importance, Kp, Vp, CTX, lo, hi, IMPORTANT, dst=(Kd, Vd, Krd), plan=plan, idx=idx,
                         rope=(cos, sin))
        if pq is not None:
            for l in range(L):
                ops.pq_pack_planes(Krd[l], Vd[l], planes[l], 0, new_len)

    def layer_step(l, n, par):                 # one layer's attention step; n = cache length AFTER the append
        ids = hid[l]
        if ids is not None and ids.numel() == 0:
            return                             # every head of this rank is pruned in this layer: its output slice stays zero
        if pq is None:
            ops.attn_decode(q[l], Kd[l], Krd[l], Vd[l], n, cos, sin, n - 1, k_new=kn[l], v_new=vn[l],
                            scores=stash[l], out=outs2[par][l], workspace=ws, head_ids=ids)
        else:
            # (r05: the step's append + plane rows inside the MSB pass; r05 earlier: spatten_kv_append_planes, one launch; r04: two)
            ops.attn_decode_pqv(q[l], planes[l], n, cos, sin, n - 1, cfg["pq_threshold"], out=outs2[par][l], need_lsb=need[l],
                                scores=stash[l], head_ids=ids, workspace=ws, append=(kn[l], vn[l], Kd[l], Krd[l], Vd[l]))

    # ---- round 6: the token's layer-steps as ONE chained launch (ops.DecodeChain) — where no collective sits between layers ----
    chains = None
    launch_mode = ["per-layer"]                # what decode_token issues (a list: flipped for the side-by-side measurement)
    # (a head-parallel run with the dependency-faithful exchange has an all-gather BETWEEN the layers: per-layer launches)
    want_chain = args.launch != "per-layer" and pq is None and not (dist_on and args.gather == "native" and args.exchange == "per-layer")
    if want_chain:
        try:
            chains = [ops.DecodeChain(q, Kd, Krd, Vd, outs2[par], k_new=kn, v_new=vn, scores=stash,
                                      head_ids=hid if cfg["head_keep"] else None) for par in range(2)]
            launch_mode[0] = "chained"
        except Exception as e:      # noqa: BLE001 - the per-layer launches are always there
            if rank == 0:
                print(f"chained launch unavailable ({type(e).__name__}: {e}); per-layer launches", file=sys.stderr)
            chains = None
    if args.launch == "chained" and chains is None and rank == 0:
        print("--launch chained: not applicable here (progressive-quant planes, or a collective between the layers); per-layer launches",
              file=sys.stderr)

    def decode_token(n, par=0):
        if launch_mode[0] == "chained":
            try:
                chains[par](n, cos, sin, n - 1)
                return
            except NotImplementedError:
                launch_mode[0] = "per-layer"
        for l in range(L):
            layer_step(l, n, par)

    def gather_token(par):
        """The exchange step of the head-parallel path: every layer's [B, H/N*d] slice -> [B, H*d] on all ranks.
        Collectives are NOT captured into HIP graphs (torch's RCCL watchdog aborts on captured work on this
        stack), so they are issued eagerly after the token's attention graph: one all-gather of the flat
        [L, B, H/N*d] buffer (default), the 32 per-layer all-gathers as one RCCL group, or one by one.  Returns an object whose ``wait()``
        orders the current stream after the collectives (no host block)."""
        import torch.distributed as dist
        if args.gather == "flat":          # [world, L, B, H/N*d]: rank r's block holds its heads of every layer
            return dist.all_gather_into_tensor(staging_flat[par], outs_flat[par].view(-1), async_op=True)
        oo, ss = outs2[par], staging2[par]
        if args.gather == "grouped" and hasattr(dist, "_coalescing_manager"):
            with dist._coalescing_manager(device=dev, async_ops=True) as cm:
                for l in range(L):
                    dist.all_gather_into_tensor(ss[l].view(world_eff * B, 1, Hl * d), oo[l].view(B, 1, Hl * d))
            return cm
        works = [dist.all_gather_into_tensor(ss[l].view(world_eff * B, 1, Hl * d), oo[l].view(B, 1, Hl * d),
                                             async_op=True) for l in range(L)]

        class _All:
            def wait(self_inner):
                for w in works:
                    w.wait()
        return _All()

    native = False
    rccl_ranks = None
    if dist_on and args.gather == "native":
        err = None
        try:
            hp.init_native()
            native = True
        except Exception as e:      # noqa: BLE001 - any failure here means "use torch.distributed instead"
            err = e
        # every rank must take the same branch: one that fell back while the others entered the native collective would hang
        ok = torch.tensor([1 if native else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if native:
                hp.close_native()
            native = False
            if rank == 0:
                print(f"native RCCL communicator unavailable ({type(err).__name__ if err else 'on another rank'}: {err}); "
                      "using torch.distributed", file=sys.stderr)
            args.gather = "flat"
        if native:
            try:
                rccl_ranks = hp.native_info()[0]
            except Exception:
                rccl_ranks = None

    peer = None
    if dist_on and native and args.peer_store:
        try:
            peer = hp.init_peer_store(max(staging2[0][0].numel() * 2 // world_eff, staging_flat[0].numel() * 2 // world_eff))
        except Exception as e:      # noqa: BLE001
            if rank == 0:
                print(f"peer-store all-gather unavailable ({type(e).__name__}: {e}); using RCCL", file=sys.stderr)
            peer = None
        okp = torch.tensor([1 if peer is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okp, op=dist.ReduceOp.MIN)
        if int(okp.item()) == 0:
            peer = None

    def exchange(send, recv):
        if peer is not None:
            hp.allgather_peer(send, recv)
        else:
            hp.allgather_native(send, recv)

    def run_slot(slot, with_exchange=True):
        if slot == 0:
            prune()
        n, par = new_len + slot + 1, slot & 1
        if native and with_exchange and args.exchange == "per-layer":
            # dependency-faithful: layer l's gathered output exists before layer l + 1 is launched (same stream, same graph)
            for l in range(L):
                layer_step(l, n, par)
                exchange(outs2[par][l].view(-1), staging2[par][l].view(-1))
            return
        decode_token(n, par)
        if native and with_exchange:      # 'flat': the exchange is part of the token — same stream, same graph — but ONE per token
            exchange(outs_flat[par].view(-1), staging_flat[par])

    # ---- HIP graphs: one per position in the turn (kv_len is a launch parameter) -------------------------
    graphs = None
    graphs_per_layer = None                    # the same slots with one launch per layer (when `graphs` holds the chained form)
    use_graph = not args.no_graph

    def capture_slots():
        for s in range(2):
            run_slot(s)
        torch.cuda.synchronize()
        gs = []
        for slot in range(TURN):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run_slot(slot)
            gs.append(g)
        for g in gs[:2]:                       # a captured collective must also replay
            g.replay()
        torch.cuda.synchronize()
        return gs
    if use_graph:
        try:
            graphs = capture_slots()
            if launch_mode[0] == "chained":
                for ch in chains:
                    ch.check()
                launch_mode[0] = "per-layer"
                try:
                    graphs_per_layer = capture_slots()
                finally:
                    launch_mode[0] = "chained"
        except Exception as e:  # e.g. a collective that cannot be captured: fall back to eager launches
            if rank == 0:
                print(f"graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()

    pending = [None, None]                    # in-flight gather per output buffer parity

    # slot of step i: the warm-up ENDS at a turn boundary, so the timed region starts with slot 0 (the prune event)
    base_slot = (-args.warmup) % TURN

    def run_steps(n, first=0, gs=None):
        gs = graphs if gs is None else gs
        for i in range(first, first + n):
            slot = (base_slot + i) % TURN
            par = slot & 1
            if dist_on and not native and pending[par] is not None:
                pending[par].wait()            # the gather that still reads this parity's outputs (token i-2)
                pending[par] = None
            if gs is not None:
                gs[slot].replay()
            else:
                run_slot(slot)
            if dist_on and not native:
                pending[par] = gather_token(par)    # overlaps the next token's attention graph
        for par in (0, 1):                     # the timed region ends with every exchange complete
            if pending[par] is not None:
                pending[par].wait()
                pending[par] = None

    run_steps(args.warmup)
    elapsed = time_region(lambda n: run_steps(n, args.warmup), args.steps, dist_on)
    tokens_per_s = B * args.steps / elapsed
    # c5: the refetch rate of the timed run, and the OTHER confidence setting beside it (same graphs: the new rows are read through
    # fixed tensors)
    pq_side = None
    if pq is not None and graphs is not None:
        torch.cuda.synchronize()
        frac_of = lambda: float(sum(float(need[l][hid[l].long()].float().mean().item()) if hid[l] is not None and hid[l].numel() else
                                    float(need[l].float().mean().item()) for l in range(L)) / L)
        pq_side = {"confidence": args.pq_confidence, "refetch_fraction": round(frac_of(), 4)}
        other = "uniform" if args.pq_confidence == "trace" else "trace"
        for l in range(L):
            kn[l].copy_(kn_uniform[l] if other == "uniform" else kn_trace[l])
        run_steps(args.warmup)
        el_o = time_region(lambda n: run_steps(n, args.warmup), args.steps, dist_on)
        torch.cuda.synchronize()
        pq_side.update({f"{other}_tokens_per_s": round(B * args.steps / el_o, 2), f"{other}_refetch_fraction": round(frac_of(), 4)})
        for l in range(L):
            kn[l].copy_(kn_trace[l] if args.pq_confidence == "trace" else kn_uniform[l])
        # ... and the same layer-steps over 16-BIT keys and values (the lean head-list step on the rotated shadow): what MSB-first buys
        try:
            gb = torch.cuda.CUDAGraph()
            nb = new_len + TURN // 2

            def bf16_token():
                for l in range(L):
                    if hid[l] is not None and hid[l].numel() == 0:
                        continue
                    ops.attn_decode(q[l], Kd[l], Krd[l], Vd[l], nb, cos, sin, nb - 1, k_new=kn[l], v_new=vn[l], scores=stash[l],
                                    out=outs2[0][l], workspace=ws, head_ids=hid[l])
            bf16_token()
            torch.cuda.synchronize()
            with torch.cuda.graph(gb):
                bf16_token()
            for _ in range(3):
                gb.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                gb.replay()
            torch.cuda.synchronize()
            us_b = (time.perf_counter() - t0) / 30 * 1e6
            pq_side["bf16_keys_same_steps_us_per_layer"] = round(us_b / L, 2)
            pq_side["bf16_keys_same_steps_tokens_per_s_decode_only"] = round(1e6 / us_b, 2)
        except Exception as e:      # noqa: BLE001
            pq_side["bf16_keys_error"] = f"{type(e).__name__}: {e}"
        prune()                     # (the 16-bit steps appended rows: restore the turn's start)
    # the same K steps with one launch per layer (the r01-r05 form) beside the chained value
    tokens_per_s_per_layer = None
    if graphs_per_layer is not None:
        run_steps(args.warmup, 0, graphs_per_layer)
        el2 = time_region(lambda n: run_steps(n, args.warmup, graphs_per_layer), args.steps, dist_on)
        tokens_per_s_per_layer = B * args.steps / el2

    # ---- what the exchange costs per token (round 4): the same decode-only slots replayed with and without the collectives,
    # as graphs, over the same number of tokens; and — native communicator — the OTHER schedule (flat / per-layer) beside the
    # one the headline number ran, so that both are on every line (also at one rank with --force-dist)
    comm = None
    if dist_on and native and graphs is not None:
        try:
            def capture(fn, slots):
                gs = []
                for sl in slots:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        fn(sl)
                    gs.append(g)
                return gs

            def time_graphs(gs, reps=4):
                for g in gs:
                    g.replay()
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    for g in gs:
                        g.replay()
                torch.cuda.synchronize()
                dt_ = torch.tensor([(time.perf_counter() - t0) / (reps * len(gs))], dtype=torch.float64, device=dev)
                dist.all_reduce(dt_, op=dist.ReduceOp.MAX)
                return float(dt_.item()) * 1e6

            slots = list(range(1, 17))                     # decode-only slots (slot 0 carries the prune event)
            run_slot(1, False); run_slot(2, False)
            torch.cuda.synchronize()
            t_none = time_graphs(capture(lambda sl: run_slot(sl, False), slots))
            t_this = time_graphs([graphs[sl] for sl in slots])
            other = "flat" if args.exchange == "per-layer" else "per-layer"
            keep = args.exchange
            args.exchange = other
            run_slot(1); run_slot(2)
            torch.cuda.synchronize()
            t_other = time_graphs(capture(run_slot, slots))
            args.exchange = keep
            comm = {"exchange": keep, "us_per_token_no_exchange": round(t_none, 1), "us_per_token": round(t_this, 1),
                    "comm_us_per_token": round(t_this - t_none, 1),
                    f"us_per_token_{other.replace('-', '_')}": round(t_other, 1),
                    f"comm_us_per_token_{other.replace('-', '_')}": round(t_other - t_none, 1),
                    "collectives_per_token": L if keep == "per-layer" else 1,
                    "bytes_per_rank_per_collective": (B * Hl * d * 2) if keep == "per-layer" else (L * B * Hl * d * 2),
                    "transport": "peer-store (hipIpc direct writes)" if peer is not None else "RCCL all-gather"}
        except Exception as e:      # noqa: BLE001 - the headline number must not depend on this side measurement
            comm = {"error": f"{type(e).__name__}: {e}"}

    result = {
        "metric": ("decode tokens/sec (attention path), Llama-2-7B N=4k, 50% token prune" if args.config in ("c2", "c3") else
                   "decode tokens/sec (attention path), Llama-2-13B N=16k, 50% token prune"),
        "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "comm": comm,
        "config": {"workload": cfg["workload"], "name": args.config,
                   "heads_launched_per_layer_this_rank": kept_heads,
                   "pq_profile": None if pq is None else {"key_msb_bits": pq[0], "value_bits": pq[1], "lsb_bits": 4,
                                                          "threshold": cfg["pq_threshold"]},
                   "layers": L, "heads": HEADS, "head_dim": d, "batch": B, "kv_len_before_prune": CTX,
                   "kv_len_after_prune": new_len, "turn_tokens": TURN,
                   "prune_events_in_timed_region": -(-args.steps // TURN),
                   "parallelism": (f"head-parallel x{world}, {args.scaling} scaling (B = {B}, H/{world} heads per rank; "
                                   f"all-gather of every layer's output, {args.gather}"
                                   f"{', ' + args.exchange if native else ''})") if dist_on else "single GPU",
                   "exchange": (args.exchange if native else args.gather) if dist_on else None,
                   "comm_us_per_token": None if not comm else comm.get("comm_us_per_token"),
                   "rccl_ranks": rccl_ranks if native else (world if dist_on else None),
                   "launch": ("hip-graph" if graphs is not None else "eager") +
                             (", chained: ONE launch per token (spatten_attn_decode_chain; a workgroup walks the layers, layer l+1's "
                              "K/V tile requested before layer l's completion is waited for; bit-identical to the per-layer launches)"
                              if launch_mode[0] == "chained" else ", one launch per layer")},
    }
    if pq_side is not None:
        pq_side["pq_us_per_layer_this_run"] = round(elapsed / args.steps * 1e6 / L, 2)
        result["config"]["pq_confidence"] = pq_side
    if tokens_per_s_per_layer is not None:
        result["per_layer_launch"] = {"tokens_per_s": round(tokens_per_s_per_layer, 2),
                                      "chained_over_per_layer": round(tokens_per_s / tokens_per_s_per_layer, 3),
                                      "note": "the same steps, same graphs-per-position, one launch per layer (the r01-r05 form)"}

    if rank == 0:
        # ---- roofline of the dominant kernel (decode attention, HBM-bound) -------------------------------
        # average launch duration, live: HIP events on the launch stream around replays of the 63 decode-only
        # slots; 32 launches per slot.  The figure includes the ~1 us gap between dependent launches, so it
        # under-states the kernel (rocprofv3 --kernel-trace gives the bare duration: profiles/).
        if graphs is not None and not dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                for slot in range(1, TURN):
                    graphs[slot].replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * (TURN - 1) * L)
            n_avg = new_len + 1 + (TURN) / 2.0
            h_act = Hl if kept_heads is None else sum(kept_heads) / float(L)   # heads launched per layer (head pruning)
            if pq is None:
                algo_bytes = 2 * B * h_act * n_avg * d * 2 + 2 * B * h_act * d * 2 + B * h_act * n_avg * 2
            else:
                # SURVEY 8d "progressive quant decode": the key MSB plane + the value plane of the kept rows (+ their fp32 row
                # scales), Q / O, the stash; a refetch adds the 4-bit LSB plane of the flagged heads (counted from need_lsb)
                n_ref = float(sum(int(x.sum().item()) for x in need)) / L
                algo_bytes = (B * h_act * n_avg * d * (pq[0] + pq[1]) / 8 + 2 * B * h_act * n_avg * 4 + 2 * B * h_act * d * 2
                              + B * h_act * n_avg * 2 + n_ref * n_avg * d / 2)
            gbs = algo_bytes / us / 1e3
            chained_now = launch_mode[0] == "chained"
            layers_per_launch = L if chained_now else 1          # the chained launch IS the token: all layers' bytes, one duration
            # HBM traffic per launch: PMC counters collected in separate rocprofv3 --pmc passes of THIS command and config
            # (tools/pmc_bench.sh: FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md), committed under profiles/ as
            # *pmc_decode_<config>.json; a file measured on the other launch form (or none for this config) gives null
            traffic = None
            pm = []
            try:
                pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"pmc_decode_{args.config}.json"))
                if pm:
                    pj_ = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))
                    # (c5: measured at the same confidence setting — the refetch pass re-streams V for the flagged heads)
                    if pj_.get("launch") == launch_mode[0] and (pq is None or pj_.get("pq_confidence") == args.pq_confidence):
                        traffic = int(pj_["traffic_bytes_per_launch"])
            except Exception:
                traffic = None
            kname = ("decode_chain_kernel<bf16,128,5,...,512> (decode_chain.hip: decode_body walked over the layers by resident "
                     "workgroups; ONE launch per token, avg_launch_us and the bytes are the whole token's)" if chained_now else
                     "decode_lean_kernel<bf16,128,5,...,512> (decode_attn.hip; 512-thread team, two waves per SIMD)" if headline else
                     "decode_lean_hids_kernel<bf16,128,5,...,512> (decode_attn.hip; the lean step over a head list)" if pq is None else
                     "pqv_decode_kernel (pq_decode.hip; the layer-step = MSB pass with the row's append + pack inside (+ LSB refetch): "
                     "avg_launch_us is the whole layer-step)")
            result["roofline"] = {"kernel": kname, "bound": "hbm", "achieved": round(gbs, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                  "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4), "achievable_GBs": HBM_ACHIEVABLE_GBS,
                                  "traffic": traffic,
                                  "traffic_source": ("committed PMC profile profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                     "of this command, tools/pmc_bench.sh) - not measured in this run" % pm[-1])
                                  if traffic is not None else None,
                                  "avg_launch_us": round(us * layers_per_launch, 3),
                                  "layers_per_launch": layers_per_launch, "us_per_layer_step": round(us, 3),
                                  "algorithmic_bytes_per_launch": int(algo_bytes * layers_per_launch)}
            if graphs_per_layer is not None:      # the per-layer launches of the same steps, same events
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    for slot in range(1, TURN):
                        graphs_per_layer[slot].replay()
                e1.record()
                torch.cuda.synchronize()
                us_pl = e0.elapsed_time(e1) * 1e3 / (reps * (TURN - 1) * L)
                result["per_layer_launch"].update({"avg_launch_us": round(us_pl, 3), "achieved_GBs": round(algo_bytes / us_pl / 1e3, 1),
                                                   "frac_of_hbm_peak": round(algo_bytes / us_pl / 1e3 / HBM_PEAK_GBS, 4)})
            # the prune event (select + fused gather): separate, informative
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                prune()
            e1.record()
            torch.cuda.synchronize()
            pus = e0.elapsed_time(e1) * 1e3 / 5
            gather_bytes = 2 * 2 * L * B * Hl * new_len * d * 2            # K,V x read+write
            moved = gather_bytes * 5 // 4                                   # + the rotated shadow written by the same pass
            # the reference's gather+concat alone (no shadow output): the "pruned KV gather" of the north star
            plan2 = ops.PrunePlan(importance, Kp, Vp, Kd, Vd, None)
            ops.prune_layers(importance, Kp, Vp, CTX, lo, hi, IMPORTANT, dst=(Kd, Vd, None), plan=plan2, idx=idx)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                ops.prune_layers(importance, Kp, Vp, CTX, lo, hi, IMPORTANT, dst=(Kd, Vd, None), plan=plan2, idx=idx)
            e1.record()
            torch.cuda.synchronize()
            gus = e0.elapsed_time(e1) * 1e3 / 5
            result["prune_event"] = {
                "us_all_layers": round(pus, 1), "bytes_moved": int(moved),
                "GBs_moved_incl_select": round(moved / pus / 1e3, 1), "frac_of_hbm_peak": round(moved / pus / 1e3 / HBM_PEAK_GBS, 4),
                "gather_only_us": round(gus, 1), "gather_only_GBs_incl_select": round(gather_bytes / gus / 1e3, 1),
                "gather_only_frac_of_hbm_peak": round(gather_bytes / gus / 1e3 / HBM_PEAK_GBS, 4)}
            if headline:
                # the same event in cascade mode (importance = fp32 accumulators of softmax probabilities, README.md:11):
                # select over fp32 scores + the fused gather + the accumulators' rows, three launches for all layers
                accs = [torch.rand(Hl, CTX, device=dev, generator=gen) for _ in range(L)]
                acc_new = torch.zeros(L, Hl, cap, dtype=torch.float32, device=dev)
                plan3 = ops.PrunePlan(accs, Kp, Vp, Kd, Vd, Krd, accs, [acc_new[l] for l in range(L)])
                casc = lambda: ops.prune_layers(accs, Kp, Vp, CTX, lo, hi, IMPORTANT, dst=(Kd, Vd, Krd), plan=plan3, idx=idx,
                                                rope=(cos, sin), acc=(accs, [acc_new[l] for l in range(L)]))
                casc()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    casc()
                e1.record()
                torch.cuda.synchronize()
                result["prune_event"]["cascade_prune_event_us"] = round(e0.elapsed_time(e1) * 1e3 / 5, 1)
                # layer-to-layer cascade (the traces' key_fetch_num shrinks layer by layer): layer l keeps k_l window tokens
                # among those layer l-1 kept, k from 1020 down to 510; one chain kernel (a workgroup per head walks the
                # layers) + ragged gathers; includes the allocation of the new planes (ops.prune_layer_cascade returns them)
                keeps_lc = [IMPORTANT - (IMPORTANT // 2) * l // (L - 1) for l in range(L)]
                lc = lambda: ops.prune_layer_cascade(importance, [None] * L, 0, Kp, Vp, [CTX] * L, [hi] * L, keeps_lc, START,
                                                     [cap] * L, (cos, sin), accs)
                lc()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    lc()
                torch.cuda.synchronize()
                result["prune_event"]["layer_cascade_prune_event_us_incl_allocation"] = round((time.perf_counter() - t0) / 3 * 1e6, 1)
                # round 4: like the plain event above, into PRE-ALLOCATED destination planes, by device events
                nl_ = [START + k_ + (CTX - hi) for k_ in keeps_lc]
                Kd_lc = [torch.empty(B, Hl, cap, d, dtype=dt, device=dev) for _ in range(L)]
                Vd_lc = [torch.empty_like(x) for x in Kd_lc]
                Krd_lc = [torch.empty_like(x) for x in Kd_lc]
                for tag_ in ("layer_cascade_prune_event_us",):
                    # like the plain event above (PrunePlan): tables and pointer rows prebuilt, the timed call is the C call alone
                    lc_plan = ops.LayerCascadePlan(importance, [None] * L, 0, Kp, Vp, [CTX] * L, [hi] * L, keeps_lc, START,
                                                   [cap] * L, (cos, sin), accs, dst=(Kd_lc, Vd_lc, Krd_lc))
                    lc2 = lc_plan.run
                    lc2()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(5):
                        lc2()
                    e1.record()
                    torch.cuda.synchronize()
                    result["prune_event"][tag_] = round(e0.elapsed_time(e1) * 1e3 / 5, 1)
                result["prune_event"]["layer_cascade_vs_plain_event"] = round(result["prune_event"]["layer_cascade_prune_event_us"] / result["prune_event"]["us_all_layers"], 3)
                del accs, acc_new, plan3, Kd_lc, Vd_lc, Krd_lc, lc_plan, lc2
            prune()   # restore the shadow planes for whatever runs next

        # ---- dense comparison legs ---------------------------------------------------------------------
        if not args.no_extras and not dist_on and headline:
            extras = {}
            gd = torch.cuda.CUDAGraph()
            so = torch.empty(B, Hl, CTX, dtype=dt, device=dev)

            def dense_token():
                for l in range(L):
                    ops.attn_decode(q[l], None, Krp[l], Vp[l], CTX, cos, sin, CTX - 1, scores=so, out=outs[l], workspace=ws)
            dense_token()
            torch.cuda.synchronize()
            with torch.cuda.graph(gd):
                dense_token()
            for _ in range(3):
                gd.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                gd.replay()
            torch.cuda.synchronize()
            extras["dense_fused_tokens_per_s"] = round(50 / (time.perf_counter() - t0), 2)
            inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d, 2, device=dev).float() / d))
            pk = [k[:, :, :CTX - 1] for k in Kp]
            pv = [v[:, :, :CTX - 1] for v in Vp]

            def eager_token():
                for l in range(L):
                    eager_pos_shift_layer(q[l][:, :, None], kn[l][:, :, None], vn[l][:, :, None], pk[l], pv[l], inv_freq)
            for _ in range(2):
                eager_token()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                eager_token()
            torch.cuda.synchronize()
            extras["dense_eager_posshift_tokens_per_s"] = round(5 / (time.perf_counter() - t0), 2)
            # "fair dense" line: torch SDPA over PRE-rotated K (no cat, no re-rotation, no stash)
            try:
                qs = [x[:, :, None].contiguous() for x in q]

                def sdpa_token():
                    for l in range(L):
                        torch.nn.functional.scaled_dot_product_attention(qs[l], Krp[l], Vp[l])
                for _ in range(3):
                    sdpa_token()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    sdpa_token()
                torch.cuda.synchronize()
                extras["dense_torch_sdpa_prerotated_tokens_per_s"] = round(20 / (time.perf_counter() - t0), 2)
            except Exception as e:
                extras["dense_torch_sdpa_error"] = f"{type(e).__name__}: {e}"
            extras["speedup_vs_dense_eager"] = round(tokens_per_s / extras["dense_eager_posshift_tokens_per_s"], 2)
            extras["speedup_vs_dense_fused"] = round(tokens_per_s / extras["dense_fused_tokens_per_s"], 2)
            try:
                extras.update(plugin_path_tokens_per_s(dev, dt))
            except Exception as e:
                extras["plugin_path_error"] = f"{type(e).__name__}: {e}"
            try:
                extras.update(dense_with_projections_tokens_per_s(dev, dt))
                for k_ in ("plugin_path_eager_tokens_per_s", "plugin_path_graph_tokens_per_s", "plugin_path_graph_fused_qkv_tokens_per_s",
                           "plugin_path_graph_fused_qkv_native_gemv_tokens_per_s"):
                    if k_ in extras:
                        extras[k_.replace("_tokens_per_s", "") + "_speedup_vs_dense_eager_with_projections"] = round(
                            extras[k_] / extras["dense_eager_with_projections_tokens_per_s"], 2)
            except Exception as e:
                extras["dense_with_projections_error"] = f"{type(e).__name__}: {e}"
            # ---- other rows of the scope table, measured on the same box (not part of `value`) -----------------
            try:
                Np = 8192                                                   # C4: causal prefill, q = N = 8192, one layer
                Qp, Kp2, Vp2 = rnd(1, HEADS, Np, d), rnd(1, HEADS, Np, d), rnd(1, HEADS, Np, d)
                cp, sp = ops.rope_table(Np, d, dt, dev)
                Krp2 = ops.rope_single(Kp2, cp, sp)
                op = torch.empty(1, Np, HEADS * d, dtype=dt, device=dev)
                # (reference numerics: both 16-bit roundings of every logit, modify_llama.py:111-113 — what a stash-producing
                #  forward runs; round 6: a forward WITHOUT a stash defaults to fp32 logits, measured below as ..._fast_numerics)
                for _ in range(3):
                    ops.attn_prefill(Qp, Krp2, Vp2, Np, cp, sp, 0, causal=True, out=op, numerics="reference")
                torch.cuda.synchronize()
                blocks = []             # median of three blocks of 8 calls (one run showed a 5x outlier block on a shared host)
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(8):
                        ops.attn_prefill(Qp, Krp2, Vp2, Np, cp, sp, 0, causal=True, out=op, numerics="reference")
                    torch.cuda.synchronize()
                    blocks.append((time.perf_counter() - t0) / 8)
                tp = sorted(blocks)[1]
                fl = 4 * HEADS * d * Np * (Np + 1) / 2
                extras["prefill_8192_causal_ms_per_layer"] = round(tp * 1e3, 3)
                extras["prefill_8192_causal_TFLOPs"] = round(fl / tp / 1e12, 1)
                extras["prefill_frac_of_bf16_mfma_peak_2500TF"] = round(fl / tp / 2.5e15, 4)
                # the run-time opt-in: fp32 logits instead of the reference's two roundings per logit (numerics="fast")
                for _ in range(3):
                    ops.attn_prefill(Qp, Krp2, Vp2, Np, cp, sp, 0, causal=True, out=op, numerics="fast")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    ops.attn_prefill(Qp, Krp2, Vp2, Np, cp, sp, 0, causal=True, out=op, numerics="fast")
                torch.cuda.synchronize()
                extras["prefill_8192_causal_fast_numerics_TFLOPs"] = round(fl / ((time.perf_counter() - t0) / 20) / 1e12, 1)
                # BASELINE.json configs[3] AS WRITTEN: the same prefill over progressively quantised keys (4-bit MSB plane first,
                # 4-bit LSB refetch for the rows whose max probability stays below the threshold) — MSB pass only, and with the
                # threshold bisected so that ~5 % of the query rows refetch
                try:
                    plq = ops.PQPlanes(1, HEADS, Np, d, dev)
                    ops.pq_pack(Krp2, plq, 0, Np)
                    needq = torch.empty(1, HEADS, Np, dtype=torch.int32, device=dev)
                    run_pq = lambda thr: ops.attn_prefill_pq(Qp, plq, Vp2, Np, cp, sp, 0, thr, causal=True, out=op, need_lsb=needq)
                    lo_t, hi_t = 0.0, 1.0
                    for _ in range(14):
                        mid = 0.5 * (lo_t + hi_t)
                        run_pq(mid)
                        frac = float(needq.float().mean().item())
                        lo_t, hi_t = (mid, hi_t) if frac < 0.05 else (lo_t, mid)
                    thr5 = 0.5 * (lo_t + hi_t)
                    for tag_, thr_ in (("msb_only", 0.0), ("refetch_5pct", thr5)):
                        for _ in range(2):
                            run_pq(thr_)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(8):
                            run_pq(thr_)
                        torch.cuda.synchronize()
                        extras[f"prefill_8192_pq_{tag_}_ms"] = round((time.perf_counter() - t0) / 8 * 1e3, 3)
                    run_pq(thr5)
                    extras["prefill_8192_pq_refetch_fraction"] = round(float(needq.float().mean().item()), 4)
                    extras["prefill_8192_pq_msb_only_TFLOPs"] = round(fl / (extras["prefill_8192_pq_msb_only_ms"] * 1e-3) / 1e12, 1)
                    del plq, needq
                except Exception as e:      # noqa: BLE001
                    extras["prefill_pq_error"] = f"{type(e).__name__}: {e}"
                # the first prompt of a C2 session: q = N = 2048 — one workgroup per CU, paired 128-row blocks (round 4)
                N2 = 2048
                c2, s2 = cp[:N2], sp[:N2]
                o2 = torch.empty(1, N2, HEADS * d, dtype=dt, device=dev)
                Q2, K2, V2 = Qp[:, :, :N2].contiguous(), Krp2[:, :, :N2].contiguous(), Vp2[:, :, :N2].contiguous()
                run2 = lambda: ops.attn_prefill(Q2, K2, V2, N2, c2, s2, 0, causal=True, out=o2, numerics="reference")
                for _ in range(3):
                    run2()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    run2()
                torch.cuda.synchronize()
                extras["prefill_2048_causal_TFLOPs"] = round(4 * HEADS * d * N2 * (N2 + 1) / 2 / ((time.perf_counter() - t0) / 20) / 1e12, 1)
                del Q2, K2, V2, o2
                # progressive-quant decode over the same 8192 keys: MSB-only vs always-refetch vs bf16 keys
                planes = ops.PQPlanes(1, HEADS, Np, d, dev)
                ops.pq_pack(Krp2, planes, 0, Np)
                q1 = Qp[:, :, -1].contiguous()
                o1 = torch.empty(1, HEADS * d, dtype=dt, device=dev)
                def _time(fn, n=20, reps=5):
                    """device time per call: n calls captured into one HIP graph (no host launch cost), replayed.
                    fn(i) gets the call index so that it can rotate over buffer copies (> the 256 MB Infinity Cache
                    in total): repeated calls on one small buffer would be served from cache, not HBM."""
                    side = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(side):
                        fn(0)
                        side.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=side):
                            for i in range(n):
                                fn(i)
                        for _ in range(2):
                            g.replay()
                        side.synchronize()
                        t = time.perf_counter()
                        for _ in range(reps):
                            g.replay()
                        side.synchronize()
                    return (time.perf_counter() - t) / (n * reps) * 1e6
                NC = 4                                     # 4 x 134 MB of K+V: every call streams from HBM
                Krc = [Krp2] + [Krp2.clone() for _ in range(NC - 1)]
                Vc = [Vp2] + [Vp2.clone() for _ in range(NC - 1)]
                plc = [planes]
                for kc_ in Krc[1:]:
                    pl_ = ops.PQPlanes(1, HEADS, Np, d, dev)
                    ops.pq_pack(kc_, pl_, 0, Np)
                    plc.append(pl_)
                extras["decode_8192_bf16_keys_us"] = round(_time(lambda i: ops.attn_decode(q1, None, Krc[i % NC], Vc[i % NC], Np, cp, sp, Np - 1, out=o1, workspace=ws)), 2)
                extras["decode_8192_pq_msb_only_us"] = round(_time(lambda i: ops.attn_decode_pq(q1, plc[i % NC], Vc[i % NC], Np, cp, sp, Np - 1, 0.0, out=o1, workspace=ws)), 2)
                extras["decode_8192_pq_refetch_all_us"] = round(_time(lambda i: ops.attn_decode_pq(q1, plc[i % NC], Vc[i % NC], Np, cp, sp, Np - 1, 2.0, out=o1, workspace=ws)), 2)
                # round 4 (ABI 4): the bit profiles — quantised VALUE plane + LSB-only refetch (spatten_attn_decode_pq); same rows
                for kb_, vb_ in ops.PQ_PROFILES:
                    ppl = []
                    for kc_, vc_ in zip(Krc, Vc):
                        pp_ = ops.PQProfilePlanes(1, HEADS, HEADS, Np, d, dev, key_bits=kb_, value_bits=vb_)
                        ops.pq_pack_planes(kc_, vc_, pp_, 0, Np)
                        ppl.append(pp_)
                    need_ = torch.zeros(HEADS, dtype=torch.int32, device=dev)
                    tag_ = f"decode_8192_pq_k{kb_}v{vb_}"
                    extras[tag_ + "_msb_only_us"] = round(_time(lambda i: ops.attn_decode_pqv(q1, ppl[i % NC], Np, cp, sp, Np - 1, 0.0, out=o1, need_lsb=need_, workspace=ws)), 2)
                    extras[tag_ + "_refetch_all_us"] = round(_time(lambda i: ops.attn_decode_pqv(q1, ppl[i % NC], Np, cp, sp, Np - 1, 2.0, out=o1, need_lsb=need_, workspace=ws)), 2)
                    del ppl
                extras["decode_8192_pq_refetch_all_vs_bf16_decode"] = round(extras["decode_8192_pq_k4v8_refetch_all_us"] / extras["decode_8192_bf16_keys_us"], 3)
                # configs[2] (C3): the pruned 2048-row cache with 25 % of the heads pruned (24 of 32 launched);
                # rotating over the 32 layers' slabs (1.1 GB)
                hid = torch.arange(0, HEADS, dtype=torch.int32, device=dev)[torch.arange(HEADS, device=dev) % 4 != 3].contiguous()
                extras["c3_decode_2048_24of32_heads_us"] = round(_time(lambda i: ops.attn_decode(
                    q[i % L], None, Krd[i % L], Vd[i % L], new_len, cos, sin, new_len - 1, out=outs[i % L], workspace=ws, head_ids=hid), n=L), 2)
                extras["c2_decode_2048_32_heads_us"] = round(_time(lambda i: ops.attn_decode(
                    q[i % L], None, Krd[i % L], Vd[i % L], new_len, cos, sin, new_len - 1, out=outs[i % L], workspace=ws), n=L), 2)
                # what ONE rank of a head-parallel run launches per layer (H/G heads of the same planes): the expected strong-scaling
                # curve of --gpus 2 / 4 / 8 before any communication, on every headline line
                prl = {}
                for hg in (16, 8, 4):
                    wsg = ops.DecodeWorkspace(1, hg, d, dev)
                    og = torch.empty(1, hg * d, dtype=dt, device=dev)
                    prl[f"c2_{hg}_heads"] = round(_time(lambda i: ops.attn_decode(
                        q[i % L][:, :hg], None, Krd[i % L][:, :hg], Vd[i % L][:, :hg], new_len, cos, sin, new_len - 1, out=og, workspace=wsg), n=L), 2)
                prl["c2_32_heads"] = extras["c2_decode_2048_32_heads_us"]
                extras["per_rank_launch_us"] = prl
                # round 6: the same per-rank shapes through the CHAINED launch (one launch per token over the rank's heads), per layer
                try:
                    prc = {}
                    for hg in (32, 16, 8, 4):
                        og = [torch.empty(1, hg * d, dtype=dt, device=dev) for _ in range(L)]
                        chg = ops.DecodeChain([x[:, :hg] for x in q], None, [x[:, :hg] for x in Krd], [x[:, :hg] for x in Vd], og)
                        prc[f"c2_{hg}_heads"] = round(_time(lambda i: chg(new_len, cos, sin, new_len - 1), n=1, reps=20) / L, 2)
                        del chg
                    extras["per_rank_chained_us_per_layer"] = prc
                except Exception as e:      # noqa: BLE001
                    extras["per_rank_chained_error"] = f"{type(e).__name__}: {e}"
                # batched decode (the C ABI takes a batch): B sequences on their own pruned 2048-row caches — the
                # launch's fixed costs (boundary, ramp, split merge) amortise over B x the bytes
                for Bb in (4, 8):
                    NCb = max(2, 640 // (Bb * 34))                      # rotate over > 256 MB of K/V
                    Kb = [rnd(Bb, HEADS, cap, d) for _ in range(NCb)]
                    Vb = [rnd(Bb, HEADS, cap, d) for _ in range(NCb)]
                    qb, ob = rnd(Bb, HEADS, d), torch.empty(Bb, HEADS * d, dtype=dt, device=dev)
                    sb_ = torch.empty(Bb, HEADS, cap, dtype=dt, device=dev)
                    wsb = ops.DecodeWorkspace(Bb, HEADS, d, dev)
                    n_b = new_len + TURN // 2
                    us_b = _time(lambda i: ops.attn_decode(qb, None, Kb[i % NCb], Vb[i % NCb], n_b, cos, sin, n_b - 1, out=ob,
                                                           scores=sb_, workspace=wsb), n=max(8, 2 * NCb))
                    by_b = 2 * Bb * HEADS * n_b * d * 2 + 2 * Bb * HEADS * d * 2 + Bb * HEADS * n_b * 2
                    extras[f"decode_2080_batch{Bb}_us"] = round(us_b, 2)
                    extras[f"decode_2080_batch{Bb}_frac_of_hbm_peak"] = round(by_b / us_b / 1e3 / HBM_PEAK_GBS, 4)
                    del Kb, Vb
                # the turn prefill of the multi-turn protocol: 64 new tokens on the 2048-row pruned cache of one layer
                # (key-split flash kernel + merge; run_spatten_llama.py:71-86 feeds every new prompt this way)
                qt = rnd(1, HEADS, 64, d)
                ot = torch.empty(1, 64, HEADS * d, dtype=dt, device=dev)
                extras["turn_prefill_64_on_2048_us_per_layer"] = round(_time(lambda i: ops.attn_prefill(
                    qt, Krd[i % L][:, :, :new_len + 64], Vd[i % L][:, :, :new_len + 64], new_len + 64, cos, sin, new_len,
                    causal=True, out=ot), n=L), 2)
                del planes, Krp2, Vp2, Qp, Krc, Vc, plc
                # configs[4] (C5): Llama-2-13B geometry (H = 40), 16384-token cache pruned to 8192 rows
                # (start 4 / important 4092 / recent 4096), one layer: decode over bf16 keys and over PQ planes
                H5, N5 = 40, 8192
                c5, s5 = ops.rope_table(2 * N5 + 64, d, dt, dev)
                q5 = rnd(1, H5, d)
                K6 = [rnd(1, H5, 2 * N5, d) for _ in range(2)]      # 2 x 671 MB dense 16384-token K+V
                V6 = [rnd(1, H5, 2 * N5, d) for _ in range(2)]
                K5 = [K6[0][:, :, :N5], K6[0][:, :, N5:], K6[1][:, :, :N5], K6[1][:, :, N5:]]   # 4 x 168 MB views
                V5 = [V6[0][:, :, :N5], V6[0][:, :, N5:], V6[1][:, :, :N5], V6[1][:, :, N5:]]
                K5 = [x.contiguous() for x in K5]
                V5 = [x.contiguous() for x in V5]
                o5 = torch.empty(1, H5 * d, dtype=dt, device=dev)
                ws5 = ops.DecodeWorkspace(1, H5, d, dev)
                pl5 = []
                for kc_ in K5:
                    pl_ = ops.PQPlanes(1, H5, N5, d, dev)
                    ops.pq_pack(kc_, pl_, 0, N5)
                    pl5.append(pl_)
                extras["c5_decode_13b_8192_kept_bf16_keys_us"] = round(_time(lambda i: ops.attn_decode(q5, None, K5[i % 4], V5[i % 4], N5, c5, s5, N5 - 1, out=o5, workspace=ws5)), 2)
                extras["c5_decode_13b_8192_kept_pq_msb_only_us"] = round(_time(lambda i: ops.attn_decode_pq(q5, pl5[i % 4], V5[i % 4], N5, c5, s5, N5 - 1, 0.0, out=o5, workspace=ws5)), 2)
                extras["c5_decode_13b_16384_dense_bf16_keys_us"] = round(_time(lambda i: ops.attn_decode(q5, None, K6[i % 2], V6[i % 2], 2 * N5, c5, s5, 2 * N5 - 1, out=o5, workspace=ws5)), 2)
                # round 4: the (8, 8) profile — the RTL harness default — on the kept 8192 rows
                pp5 = []
                for kc_, vc_ in zip(K5, V5):
                    pp_ = ops.PQProfilePlanes(1, H5, H5, N5, d, dev, key_bits=8, value_bits=8)
                    ops.pq_pack_planes(kc_, vc_, pp_, 0, N5)
                    pp5.append(pp_)
                need5 = torch.zeros(H5, dtype=torch.int32, device=dev)
                extras["c5_decode_13b_8192_kept_pq_k8v8_msb_only_us"] = round(_time(lambda i: ops.attn_decode_pqv(q5, pp5[i % 4], N5, c5, s5, N5 - 1, 0.0, out=o5, need_lsb=need5, workspace=ws5)), 2)
                del pp5, pl5
                # one rank's share of that launch at --gpus 2 / 4 / 8 (20 / 10 / 5 of the 40 heads, k8v8 MSB pass)
                prl["c5_40_heads_pq_k8v8"] = extras["c5_decode_13b_8192_kept_pq_k8v8_msb_only_us"]
                for hg in (20, 10, 5):
                    ppg = []
                    for kc_, vc_ in zip(K5, V5):
                        pp_ = ops.PQProfilePlanes(1, hg, hg, N5, d, dev, key_bits=8, value_bits=8)
                        ops.pq_pack_planes(kc_[:, :hg], vc_[:, :hg], pp_, 0, N5)
                        ppg.append(pp_)
                    wsg = ops.DecodeWorkspace(1, hg, d, dev)
                    og = torch.empty(1, hg * d, dtype=dt, device=dev)
                    needg = torch.zeros(hg, dtype=torch.int32, device=dev)
                    qg = q5[:, :hg].contiguous()
                    prl[f"c5_{hg}_heads_pq_k8v8"] = round(_time(lambda i: ops.attn_decode_pqv(qg, ppg[i % 4], N5, c5, s5, N5 - 1, 0.0, out=og, need_lsb=needg, workspace=wsg)), 2)
                    del ppg
                # round 4: local V pruning as ONE launch on the dense 16384-row cache, 30 % of the V rows fetched
                # (SpAttenController.scala:546-558,591-612) beside the plain step over the same rows (above)
                st5 = torch.empty(1, H5, 2 * N5, dtype=dt, device=dev)
                keep5 = int(0.3 * 2 * N5)
                extras["c5_decode_13b_16384_local_v_30pct_us"] = round(_time(lambda i: ops.attn_decode_local_v(
                    q5, K6[i % 2], V6[i % 2], 2 * N5, c5, s5, 2 * N5 - 1, keep5, st5, out=o5)), 2)
                extras["c5_local_v_30pct_vs_plain_decode"] = round(extras["c5_decode_13b_16384_local_v_30pct_us"] / extras["c5_decode_13b_16384_dense_bf16_keys_us"], 3)
                # ... and as the plugin runs the step: WITH the append of the token's row (inside the launch since round 5), beside the
                # plain decode step with ITS append (always inside the launch)
                kn5 = torch.randn(1, H5, d, device=dev, dtype=torch.float32).to(dt)
                vn5 = torch.randn(1, H5, d, device=dev, dtype=torch.float32).to(dt)
                ku6 = torch.zeros_like(K6[0])             # the un-rotated plane both steps append to
                extras["c5_decode_13b_16384_dense_with_append_us"] = round(_time(lambda i: ops.attn_decode(
                    q5, ku6, K6[i % 2], V6[i % 2], 2 * N5, c5, s5, 2 * N5 - 1, k_new=kn5, v_new=vn5, out=o5, workspace=ws5)), 2)
                extras["c5_decode_13b_16384_local_v_30pct_with_append_us"] = round(_time(lambda i: ops.attn_decode_local_v(
                    q5, K6[i % 2], V6[i % 2], 2 * N5, c5, s5, 2 * N5 - 1, keep5, st5, out=o5, k_new=kn5, v_new=vn5, k_cache=ku6)), 2)
                extras["c5_local_v_30pct_vs_plain_decode_with_append"] = round(
                    extras["c5_decode_13b_16384_local_v_30pct_with_append_us"] / extras["c5_decode_13b_16384_dense_with_append_us"], 3)
                del K6, V6, K5, V5, ku6, st5
                # round 6: a grouped-query model's step (32 query heads on 8 kv heads, modify_llama.py:106-108 repeat_kv; not among
                # BASELINE's configs): one workgroup column per query head against a kv head's rows streamed once and scored for the
                # whole group on the matrix cores (csrc/decode_gqa.hip); "unique" = K/V bytes counted once
                Hq, Hk = 32, 8
                for Ng in (16384, 4096):
                    capg = Ng + 64
                    cg, sg = ops.rope_table(capg + 8, d, dt, dev)
                    Lg = max(4, int(600e6 // (2 * Hk * capg * d * 2)) + 1)
                    Kg = [rnd(1, Hk, capg, d) for _ in range(Lg)]
                    Vg = [rnd(1, Hk, capg, d) for _ in range(Lg)]
                    Kug = torch.zeros_like(Kg[0])
                    qg_, kng, vng = rnd(1, Hq, d), rnd(1, Hk, d), rnd(1, Hk, d)
                    og_ = torch.empty(1, Hq * d, dtype=dt, device=dev)
                    wsg_ = ops.DecodeWorkspace(1, Hq, d, dev)
                    prev_mode = ops.set_decode_gqa(-1)
                    try:
                        for mode, tag in ((0, "per_query_head"), (1, "matrix_core"), (-1, "default")):
                            ops.set_decode_gqa(mode)
                            extras[f"gqa_32over8_{Ng}_{tag}_us"] = round(_time(lambda i: ops.attn_decode(
                                qg_, Kug, Kg[i % Lg], Vg[i % Lg], Ng, cg, sg, Ng - 1, k_new=kng, v_new=vng, out=og_, workspace=wsg_), n=Lg), 2)
                    finally:
                        ops.set_decode_gqa(prev_mode)
                    if Ng == 16384:
                        extras["gqa_32over8_16384_matrix_core_unique_frac_of_hbm_peak"] = round(
                            2 * Hk * Ng * d * 2 / (extras["gqa_32over8_16384_matrix_core_us"] * 1e-6) / 8e12, 3)
                    del Kg, Vg, Kug
            except Exception as e:  # the headline number must not depend on the side measurements
                extras["side_measurements_error"] = f"{type(e).__name__}: {e}"
            result["extras"] = extras

        # ---- CPU baseline: the reference's op sequence on the host cores, bounded sample -------------------------------
        # `value` = the torch-CPU mirror of the reference path (oracle/torch_mirror.py: the same eager torch ops the
        # reference's Python issues, timed within +-20 % of the imported reference in the build container —
        # profiles/r02_cpu_port_vs_reference.json) at all physical cores; beside it the same at one thread and the C
        # port (oracle/oracle.c, OpenMP over heads).
        if not args.no_cpu_baseline and not dist_on and headline:
            result["cpu_baseline"] = cpu_baseline(L, new_len, lo, hi)
    if dist_on:
        import torch.distributed as dist
        if native:
            hp.close_native()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its banner through C stdio: flush it first so the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
