#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (`/root/reference/spatten_llm`).

Run in the build container only (the reference does not travel to the GPU box):

    python tests/golden/gen_golden.py

Writes ``tests/golden/*.npz``: expected OUTPUTS of the reference on inputs that
are regenerated deterministically from ``oracle.spatten_oracle.synth_normal``
(seed / tensor-id / shape recorded in each case), plus an input checksum so a
drifting generator is detected.  Fixtures are data only — no reference source.

Reference entry points driven (SURVEY §8c recipe):
  * spatten_llm.pos_shift.modify_llama.apply_rotary_pos_emb_single      (G4)
  * spatten_llm.pos_shift.modify_llama.llama_pos_shift_attention_forward (G3) through a
    stub nn.Module exposing the transformers-4.33 attribute surface
  * spatten_llm.kv_cache_token_pruning.SpAttenKVCache.apply_token_pruning (G1, G2, G5)
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# NOTE: the repo root must NOT be on sys.path here: its drop-in `spatten_llm/` shim is a regular
# package and would shadow the reference's namespace package of the same name.
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
sys.path.insert(0, "/root/reference")

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("spatten_oracle", os.path.join(ROOT, "oracle", "spatten_oracle.py"))
_orc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_orc)
synth_normal = _orc.synth_normal  # input generator only
from spatten_llm.kv_cache_token_pruning import SpAttenKVCache  # noqa: E402  (REFERENCE)
from spatten_llm.pos_shift.modify_llama import (  # noqa: E402  (REFERENCE)
    apply_rotary_pos_emb_single,
    llama_pos_shift_attention_forward,
)

import spatten_llm.kv_cache_token_pruning as _refmod  # noqa: E402

assert _refmod.__file__.startswith("/root/reference/"), "must import the REFERENCE spatten_llm"

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def tt(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(TORCH_DT[dt])


def nn_(t):
    return t.detach().to(torch.float32).cpu().numpy()


def cksum(*arrs):
    s = 0.0
    for a in arrs:
        s += float(np.asarray(a, dtype=np.float64).sum())
    return np.float64(s)


class Rotary433(nn.Module):
    """transformers==4.33.0 LlamaRotaryEmbedding, restated (not under /root/reference)."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        self.dim, self.base = dim, base

    def forward(self, x, seq_len=None):
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2).float() / self.dim))
        t = torch.arange(seq_len, dtype=inv_freq.dtype)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos()[None, None, :, :].to(dtype=x.dtype), emb.sin()[None, None, :, :].to(dtype=x.dtype))


class Fixed(nn.Module):
    def __init__(self):
        super().__init__()
        self.t = None

    def forward(self, x):
        return self.t


class StubAttn(nn.Module):
    def __init__(self, H, Hkv, d):
        super().__init__()
        self.config = SimpleNamespace(pretraining_tp=1)
        self.num_heads, self.num_key_value_heads, self.head_dim = H, Hkv, d
        self.num_key_value_groups = H // Hkv
        self.hidden_size = H * d
        self.q_proj, self.k_proj, self.v_proj = Fixed(), Fixed(), Fixed()
        self.o_proj = nn.Identity()
        self.rotary_emb = Rotary433(d)


def hf_causal_mask(B, ql, N, dtype):
    P = N - ql
    i = torch.arange(ql)[:, None]
    j = torch.arange(N)[None, :]
    m = torch.where(j <= P + i, torch.tensor(0.0), torch.tensor(torch.finfo(dtype).min)).to(dtype)
    return m[None, None].expand(B, 1, ql, N).contiguous()


def ref_forward(q, k, v, past, pos, mask, dt, H, Hkv, d):
    """q [B,H,ql,d] etc. (numpy, model-dtype representable).  Drives the REFERENCE forward."""
    B, _, ql, _ = q.shape
    m = StubAttn(H, Hkv, d)
    m.q_proj.t = tt(q, dt).transpose(1, 2).reshape(B, ql, H * d)
    m.k_proj.t = tt(k, dt).transpose(1, 2).reshape(B, ql, Hkv * d)
    m.v_proj.t = tt(v, dt).transpose(1, 2).reshape(B, ql, Hkv * d)
    hidden = torch.zeros(B, ql, H * d, dtype=TORCH_DT[dt])
    pkv = None if past is None else (tt(past[0], dt), tt(past[1], dt))
    with torch.no_grad():
        out, _, new_past = llama_pos_shift_attention_forward(
            m, hidden, attention_mask=mask, position_ids=torch.from_numpy(pos).long(),
            past_key_value=pkv, use_cache=True)
    return nn_(out), nn_(m.attn_scores), nn_(new_past[0]), nn_(new_past[1])


def gen_rope(out):
    cases = {}
    for dt in ("f32", "bf16", "f16"):
        seed = 40
        x = synth_normal(seed, 0, (2, 4, 33, 64), dt)
        rot = Rotary433(64)
        cos, sin = rot(tt(x, dt), seq_len=200)
        pos = np.stack([np.arange(33) * 5 % 200, (np.arange(33) * 7 + 3) % 200]).astype(np.int64)
        y = apply_rotary_pos_emb_single(tt(x, dt), cos, sin, torch.from_numpy(pos))
        cases[f"rope_{dt}_y"] = nn_(y)
        cases[f"rope_{dt}_cos"] = nn_(cos)[0, 0]
        cases[f"rope_{dt}_sin"] = nn_(sin)[0, 0]
        cases[f"rope_{dt}_pos"] = pos
        cases[f"rope_{dt}_inck"] = cksum(x)
        # d=128 table rows used by the Llama configs (spot rows of a 4096-row table)
        cos128, sin128 = Rotary433(128)(tt(x, dt), seq_len=4096)
        rows = np.array([0, 1, 2, 63, 64, 1000, 2047, 2048, 4095])
        cases[f"rope_{dt}_rows128"] = rows
        cases[f"rope_{dt}_cos128"] = nn_(cos128)[0, 0][rows]
        cases[f"rope_{dt}_sin128"] = nn_(sin128)[0, 0][rows]
    np.savez_compressed(os.path.join(out, "g4_rope.npz"), **cases)


ATTN_CASES = [
    # name, B, H, Hkv, d, P, q, mask, dt, seed
    ("dec_f32_p127", 1, 4, 4, 64, 127, 1, "zeros", "f32", 31),
    ("dec_f32_p127_nomask", 1, 4, 4, 64, 127, 1, None, "f32", 32),
    ("pre_f32_p0_q16", 1, 4, 4, 64, 0, 16, "causal", "f32", 33),
    ("pre_f32_p127_q16", 1, 4, 4, 64, 127, 16, "causal", "f32", 34),
    ("pre_f32_p0_q16_nomask", 1, 4, 4, 64, 0, 16, None, "f32", 35),
    ("dec_bf16_p127", 1, 4, 4, 64, 127, 1, "zeros", "bf16", 36),
    ("dec_f16_p127", 1, 4, 4, 64, 127, 1, "zeros", "f16", 37),
    ("pre_bf16_p127_q16", 1, 4, 4, 64, 127, 16, "causal", "bf16", 38),
    ("pre_f16_p0_q16", 1, 4, 4, 64, 0, 16, "causal", "f16", 39),
    ("dec_bf16_d128_p1023", 1, 8, 8, 128, 1023, 1, "zeros", "bf16", 41),
    ("dec_f16_d128_p1023", 1, 8, 8, 128, 1023, 1, "zeros", "f16", 42),
    ("dec_f32_d128_p300", 2, 8, 8, 128, 300, 1, "zeros", "f32", 43),
    ("dec_bf16_gqa_p200", 1, 8, 2, 128, 200, 1, "zeros", "bf16", 44),
    ("pre_bf16_d128_p0_q128", 1, 4, 4, 128, 0, 128, "causal", "bf16", 45),
    ("pre_bf16_d128_p200_q72", 1, 4, 4, 128, 200, 72, "causal", "bf16", 46),
    ("c1_gpt2small_f32", 1, 2, 2, 64, 127, 1, "zeros", "f32", 1),   # C1 run with H=2 (A15 quirk), head 0 = the 1-head answer
    ("dec_bf16_posq_shift", 1, 4, 4, 128, 90, 1, "zeros", "bf16", 47),  # position_ids = P (HF), after-prune style short cache
]


def gen_attention(out):
    cases = {}
    meta = []
    for name, B, H, Hkv, d, P, ql, mask_kind, dt, seed in ATTN_CASES:
        kscale = 1.0
        q = synth_normal(seed, 0, (B, H, ql, d), dt)
        k = synth_normal(seed, 1, (B, Hkv, ql, d), dt, kscale)
        v = synth_normal(seed, 2, (B, Hkv, ql, d), dt)
        past = None
        if P > 0:
            past = (synth_normal(seed, 3, (B, Hkv, P, d), dt, kscale), synth_normal(seed, 4, (B, Hkv, P, d), dt))
        N = P + ql
        pos = np.tile(np.arange(P, N)[None], (B, 1)).astype(np.int64)
        if mask_kind is None:
            mask = None
        elif mask_kind == "zeros":
            mask = torch.zeros(B, 1, ql, N, dtype=TORCH_DT[dt])
        else:
            mask = hf_causal_mask(B, ql, N, TORCH_DT[dt])
        o, stash, kc, vc = ref_forward(q, k, v, past, pos, mask, dt, H, Hkv, d)
        cases[f"{name}_out"] = o
        cases[f"{name}_stash"] = stash
        cases[f"{name}_kck"] = cksum(kc)
        cases[f"{name}_vck"] = cksum(vc)
        cases[f"{name}_inck"] = cksum(q, k, v) + (0.0 if past is None else cksum(*past))
        meta.append(f"{name}|{B}|{H}|{Hkv}|{d}|{P}|{ql}|{mask_kind}|{dt}|{seed}")
    cases["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(out, "g3_attention.npz"), **cases)


PRUNE_CASES = [
    # name, H, L, d, start, recent, important, num_coming, q_stash, dt, seed
    ("p_h2_l128", 2, 128, 64, 4, 32, 28, 0, 1, "f32", 1),          # C1 params (H=2, head 0 = 1-head answer)
    ("p_h8_l300", 8, 300, 16, 4, 64, 50, 20, 1, "f32", 11),        # SURVEY appendix B probe
    ("p_h8_l300_c_gt_r", 8, 300, 16, 4, 64, 50, 100, 1, "f32", 12),  # num_coming > recent: empty tail, window clamps
    ("p_h4_l1024", 4, 1024, 64, 4, 256, 252, 64, 1, "f32", 13),
    ("p_h4_l1024_q5", 4, 1024, 64, 4, 256, 252, 0, 5, "f32", 14),   # prefill-style stash (q>1): importance sums rows
    ("p_h4_l300_s0", 4, 300, 16, 0, 100, 100, 30, 1, "f32", 15),    # start_size 0 (demo default)
    ("p_h8_l300_bf16", 8, 300, 16, 4, 64, 50, 20, 1, "bf16", 16),   # 16-bit KV payload, scores tie-free checked
    ("p_h8_l300_f16", 8, 300, 16, 4, 64, 50, 20, 1, "f16", 17),
    ("p_h3_l257_odd", 3, 257, 24, 3, 31, 77, 5, 2, "f32", 18),      # ragged sizes
]


def tie_free(score, lo, hi, k):
    """True when, per head, the k-th and (k+1)-th largest candidates differ."""
    for h in range(score.shape[0]):
        c = np.sort(score[h, lo:hi])[::-1]
        if k < c.size and c[k - 1] == c[k]:
            return False
    return True


def ref_prune(K, V, stash, start, recent, important, num_coming, dt):
    with contextlib.redirect_stdout(io.StringIO()):
        cache = SpAttenKVCache(start_size=start, recent_size=recent, important_size=important)
    past = [(tt(K, dt), tt(V, dt))]
    out = cache.apply_token_pruning(past, num_coming, [tt(stash, dt)])
    return cache, out


def gen_prune(out):
    cases = {}
    meta = []
    for name, H, L, d, start, recent, important, c, qs, dt, seed in PRUNE_CASES:
        bump = 0
        while True:
            stash = synth_normal(seed + 1000 * bump, 5, (1, H, qs, L), dt)
            with contextlib.redirect_stdout(io.StringIO()):
                sc = nn_(tt(stash, dt).sum(0).sum(1))
            if tie_free(sc, start, min(L - recent + c, L), important):
                break
            bump += 1
        K = synth_normal(seed, 6, (1, H, L, d), dt)
        V = synth_normal(seed, 7, (1, H, L, d), dt)
        cache, res = ref_prune(K, V, stash, start, recent, important, c, dt)
        Kn, Vn = nn_(res[0][0]), nn_(res[0][1])
        cases[f"{name}_imp"] = nn_(cache.importance_score[0])
        cases[f"{name}_K"] = Kn
        cases[f"{name}_V"] = Vn
        # indices recovered from the reference's own output (K rows are unique w.p. 1)
        meta.append(f"{name}|{H}|{L}|{d}|{start}|{recent}|{important}|{c}|{qs}|{dt}|{seed}|{bump}")
    cases["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(out, "g1_prune.npz"), **cases)


def gen_prune_c2(out):
    """G2: C2 scale (H=32, L=4096, start 4 / important 1020 / recent 1024) indices only.
    The reference's kept indices are read back from a K whose row j of head h encodes j."""
    cases = {}
    for tag, c, seed in (("c0", 0, 2), ("c64", 64, 3)):
        H, L, d = 32, 4096, 8
        bump = 0
        while True:
            stash = synth_normal(seed + 1000 * bump, 5, (1, H, 1, L), "f32")
            if tie_free(stash[0, :, 0], 4, min(L - 1024 + c, L), 1020):
                break
            bump += 1
        K = np.broadcast_to(np.arange(L, dtype=np.float32)[None, None, :, None], (1, H, L, d)).copy()
        _, res = ref_prune(K, K, stash, 4, 1024, 1020, c, "f32")
        kept = nn_(res[0][0])[0, :, :, 0].astype(np.uint16)      # [H, L'] absolute source positions
        cases[f"{tag}_kept"] = kept
        cases[f"{tag}_meta"] = np.array([H, L, 4, 1024, 1020, c, seed, bump])
    np.savez_compressed(os.path.join(out, "g2_prune_c2.npz"), **cases)


def gen_protocol(out):
    """G5: caller protocol (run_spatten_llama.py:60-87): multi-turn L trajectory + pruned counters,
    including passthrough turns (same object returned) and the None case."""
    H, d, dt = 4, 16, "f32"
    start, recent, important = 4, 48, 44
    with contextlib.redirect_stdout(io.StringIO()):
        cache = SpAttenKVCache(start_size=start, recent_size=recent, important_size=important)
    assert cache.apply_token_pruning(None, 10, []) is None
    L = 0
    past = None
    traj, kept_rows = [], []
    turns = [(30, 20), (25, 20), (40, 20), (10, 20), (70, 20), (33, 20)]   # (prompt_len, generated)
    for turn, (plen, gen) in enumerate(turns):
        space_needed = plen + 20
        if past is not None:
            Lp = past[0][0].shape[2]
            stash = synth_normal(50 + turn, 5, (1, H, 1, Lp), dt)
            new = cache.apply_token_pruning(past, space_needed, [tt(stash, dt)])
            same = new is past
            past = new
            traj.append([turn, Lp, space_needed, past[0][0].shape[2], int(same)])
            kept_rows.append(nn_(past[0][0])[0, :, :, 0].astype(np.int64))
        # "generate": append plen+gen rows whose payload encodes a unique global id
        Lp = 0 if past is None else past[0][0].shape[2]
        n_new = plen + gen
        ids = (1000 * (turn + 1) + np.arange(n_new)).astype(np.float32)
        new_rows = np.broadcast_to(ids[None, None, :, None], (1, H, n_new, d)).copy()
        if past is None:
            past = [(tt(new_rows, dt), tt(new_rows, dt))]
        else:
            past = [(torch.cat([past[0][0], tt(new_rows, dt)], 2), torch.cat([past[0][1], tt(new_rows, dt)], 2))]
    cases = {"traj": np.array(traj), "params": np.array([H, d, start, recent, important])}
    for i, kr in enumerate(kept_rows):
        cases[f"kept_{i}"] = kr
    cases["turns"] = np.array(turns)
    np.savez_compressed(os.path.join(out, "g5_protocol.npz"), **cases)


def bf16_bits(a):
    """fp32 array holding bf16-representable values -> their 16-bit patterns (half the bytes of the fixture)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    assert not (u & np.uint32(0xFFFF)).any()
    return (u >> np.uint32(16)).astype(np.uint16)


def gen_fullsize(out):
    """G6: the BASELINE.json geometries through the reference forward itself — C2 (Llama-2-7B: H = 32, d = 128, one decode
    step on a 4095-token cache) and C5 (Llama-2-13B: H = 40, 16383-token cache), bf16 — and the C2 prune event
    (start 4 / important 1020 / recent 1024) driven by the bf16 stash that forward produced.  Outputs and stashes are kept
    as bf16 bit patterns (C5: the stash rows of heads 0, 5, ..., 35); the kept positions are read back from a cache whose
    row j encodes j.  NOTE on ties: with 3068 bf16 candidates per head about two scores share every bf16 value near the
    1020-th largest, so a tie AT the threshold is the normal case at this scale (no seed avoids it in all 32 heads);
    torch.topk's choice among equal scores is unspecified, the HIP path keeps the lowest positions — consumers compare
    under the tie rule: everything above the threshold value identical, the same COUNT of threshold-valued positions."""
    cases = {}
    dt, d = "bf16", 128
    for name, H, P, seed0 in (("c2", 32, 4095, 61), ("c5", 40, 16383, 62)):
        seed = seed0
        q = synth_normal(seed, 0, (1, H, 1, d), dt)
        k = synth_normal(seed, 1, (1, H, 1, d), dt)
        v = synth_normal(seed, 2, (1, H, 1, d), dt)
        past = (synth_normal(seed, 3, (1, H, P, d), dt), synth_normal(seed, 4, (1, H, P, d), dt))
        N = P + 1
        pos = np.full((1, 1), P, np.int64)
        mask = torch.zeros(1, 1, 1, N, dtype=TORCH_DT[dt])
        o, stash, kc, vc = ref_forward(q, k, v, past, pos, mask, dt, H, H, d)
        cases[f"{name}_meta"] = np.array([H, P, d, seed])
        cases[f"{name}_out"] = bf16_bits(o)
        cases[f"{name}_stash"] = bf16_bits(stash if name == "c2" else stash[:, ::5])
        cases[f"{name}_inck"] = cksum(q, k, v, *past)
        if name == "c2":
            L = N
            ids = np.broadcast_to(np.arange(L, dtype=np.float32)[None, None, :, None], (1, H, L, 8)).copy()
            with contextlib.redirect_stdout(io.StringIO()):
                cache = SpAttenKVCache(start_size=4, recent_size=1024, important_size=1020)
            res = cache.apply_token_pruning([(torch.from_numpy(ids), torch.from_numpy(ids))], 0, [tt(stash, dt)])
            cases["c2_kept"] = nn_(res[0][0])[0, :, :, 0].astype(np.uint16)           # [32, 2048] source positions
            cases["c2_tied_heads"] = np.array([0 if tie_free(stash[0, h:h + 1, 0], 4, N - 1024, 1020) else 1 for h in range(H)])
    np.savez_compressed(os.path.join(out, "g6_fullsize.npz"), **cases)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "g6":
        gen_fullsize(HERE)
        return
    gen_rope(HERE)
    gen_attention(HERE)
    gen_prune(HERE)
    gen_prune_c2(HERE)
    gen_protocol(HERE)
    gen_fullsize(HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
