"""Randomised eager-vs-DecodeGraph comparison of the patched stack (developer tool): random geometry, dtype, batch and
extension modes; every hidden state, the prune that follows and a second turn must agree bit for bit.
    python tools/fuzz_graph.py [n_cases] [seed]"""
import contextlib
import io
import os
import random
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_graph_decode as T  # noqa: E402
from spatten_amd import enable_spatten_llm  # noqa: E402
from spatten_amd.graph import DecodeGraph  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    H = rng.choice([4, 8, 12])
    Hkv = rng.choice([h for h in (1, 2, 4, H) if H % h == 0 and h <= H])
    d = rng.choice([64, 128])
    dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    B = rng.choice([1, 1, 2])
    layers = rng.choice([2, 3])
    imp, rec = rng.randint(30, 60), rng.randint(30, 64)
    kw = {}
    if rng.random() < 0.4:
        kw["importance_mode"] = "cascade"
    if rng.random() < 0.35:
        kw["head_keep"] = max(1, H - rng.randint(1, 2))
    r = rng.random()
    if r < 0.3:
        kw["pq_threshold"] = rng.choice([0.02, 0.05, 0.2])
    elif r < 0.5 and "importance_mode" not in kw or (r < 0.5 and Hkv == H):
        ks = sorted([rng.randint(imp // 2, imp) for _ in range(layers)], reverse=True)
        if not ("importance_mode" in kw and Hkv != H):
            kw["layer_keep"] = ks
    if rng.random() < 0.4:
        kw["fuse_qkv"] = True
    if rng.random() < 0.4 and dt != torch.float32:
        kw["native_gemv"] = True
    P, Tn, coming = rng.randint(imp + rec + 20, 400), rng.randint(4, 9), rng.randint(6, 20)
    T.LAYERS, T.H, T.D, T.HID = layers, H, d, H * d
    tag = f"case {case}: B={B} H={H} Hkv={Hkv} d={d} {str(dt)[6:]} L={layers} imp={imp} rec={rec} P={P} T={Tn} {kw}"
    try:
        torch.manual_seed(case)

        def make():
            st = T.Stack(dt)
            for m in st.layers:
                m.num_key_value_heads, m.num_key_value_groups = Hkv, H // Hkv
                m.k_proj = nn.Linear(H * d, Hkv * d, bias=False, dtype=dt, device="cuda")
                m.v_proj = nn.Linear(H * d, Hkv * d, bias=False, dtype=dt, device="cuda")
            return st
        a = make()
        for p_ in a.parameters():
            p_.data.mul_(0.5)
        b = make()
        b.load_state_dict(a.state_dict())
        caches = []
        for m in (a, b):
            with contextlib.redirect_stdout(io.StringIO()):
                caches.append(enable_spatten_llm(m, 4, imp, rec, **kw))
        g = torch.Generator(device="cuda").manual_seed(case)
        x0 = torch.randn(B, P, H * d, device="cuda", generator=g).to(dt)
        _, pa = a(x0, None)
        _, pb = b(x0, None)
        for turn in range(2):
            graph = DecodeGraph(lambda past, x: tuple(reversed(b(x, past))), pb, horizon=Tn)
            for t in range(Tn):
                x = torch.randn(B, 1, H * d, device="cuda", generator=g).to(dt)
                ya, pa = a(x, pa)
                yb = graph.step(x)
                assert torch.equal(ya, yb), ("hidden", turn, t)
            pb = graph.past_key_values
            na = caches[0].apply_token_pruning(pa, coming, [m.attn_scores for m in a.layers])
            nb = caches[1].apply_token_pruning(pb, coming, [m.attn_scores for m in b.layers])
            ext = getattr(caches[0], "ext", None)
            kept = [slice(None) if ext is None or st.head_ids is None else None for st in (ext.layers if ext else a.layers)]
            for i, ((ka, va), (kb, vb)) in enumerate(zip(na, nb)):
                assert ka.shape == kb.shape, ("shape", turn, i)
                if kept[i] is not None or Hkv == H:
                    hk = slice(None) if kept[i] is not None else ext.layers[i].head_ids.long()
                    assert torch.equal(ka[:, hk], kb[:, hk]) and torch.equal(va[:, hk], vb[:, hk]), ("cache", turn, i)
            xp = torch.randn(B, rng.randint(3, 12), H * d, device="cuda", generator=g).to(dt)
            ya, pa = a(xp, na)
            yb, pb = b(xp, nb)
            assert torch.equal(ya, yb), ("prefill", turn)
        print("ok  ", tag, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", tag, "->", type(e).__name__, str(e)[:300], flush=True)
print(f"{n_cases - bad} / {n_cases} cases agree")
