"""Randomised sessions of the reference's caller loop (keyword model(...) calls, host read per token, prune at the turn
boundary) with and without enable_spatten_llm(auto_graph=<horizon>): tokens, logits and caches must agree bit for bit.
    python tools/fuzz_auto_graph.py [n_sessions] [seed]"""
import contextlib
import io
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_graph_decode as T  # noqa: E402
from spatten_amd import enable_spatten_llm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    dt = rng.choice([torch.bfloat16, torch.float16])
    kw = {}
    if rng.random() < 0.4:
        kw["importance_mode"] = "cascade"
    if rng.random() < 0.3:
        kw["head_keep"] = 6
    if rng.random() < 0.3:
        kw["pq_threshold"] = 0.05
        if rng.random() < 0.6 and "importance_mode" not in kw:      # (round 5: the profiled planes — append inside the MSB pass)
            kw["pq_profile"] = rng.choice([(4, 8), (8, 8), (6, 6)])
    elif rng.random() < 0.25 and "importance_mode" not in kw:        # (round 5: local V pruning — append inside the launch; with the
                                                                     #  cascade accumulation its step is not graph-capable by design)
        kw["local_v_keep"] = rng.choice([0.3, 0.6])
    if rng.random() < 0.4:
        kw.update(fuse_qkv=True, native_gemv=True)
    # (horizons that keep the slab capacity of the eager run — capacities are rounded to 128 rows: the split-N layout follows the
    #  capacity, so a graph bound to LARGER slabs adds its partials in another order and agrees to rounding, not bit for bit)
    horizon = rng.choice([3, 8, 32, 200, True]) if os.environ.get("FUZZ_ANY_HORIZON") else rng.choice([3, 8, 16])
    imp, rec = rng.randint(30, 50), rng.randint(30, 50)
    turns = [(rng.randint(20, 150), rng.randint(3, 20)) for _ in range(rng.randint(2, 4))]
    if turns[0][0] + turns[0][1] < 4 + imp + rec + 2:      # the first prune needs a window of at least `imp` candidates (the
        turns[0] = (4 + imp + rec + 2, turns[0][1])        # reference's torch.topk raises on a shorter one, and so does the cache)
    tag = f"session {case}: {str(dt)[6:]} horizon={horizon} imp={imp} rec={rec} turns={turns} {kw}"
    try:
        torch.manual_seed(case)
        a, b = T._HFStyleLM(dt), T._HFStyleLM(dt)
        b.load_state_dict(a.state_dict())
        caches = []
        for m, auto in ((a, False), (b, horizon)):
            with contextlib.redirect_stdout(io.StringIO()):
                caches.append(enable_spatten_llm(m, 4, imp, rec, auto_graph=auto, **kw))
        g = torch.Generator(device="cuda").manual_seed(case)
        prompts = [torch.randint(0, T._TinyLM.VOCAB, (1, p), device="cuda", generator=g) for p, _ in turns]

        hz = 64 if horizon is True else int(horizon)

        def session(model, cache, reserve=False):
            past, trace = None, []
            with torch.no_grad():
                for idx, ids in enumerate(prompts):
                    gen = turns[idx][1]
                    if idx > 0:
                        scores = [m.self_attn.attn_scores for m in model.layers]
                        past = cache.apply_token_pruning(past, ids.shape[1] + gen, scores)
                    o = model(input_ids=ids, past_key_values=past, use_cache=True)
                    past, tok = o.past_key_values, o.logits[:, -1, :].argmax(-1).unsqueeze(1)
                    if reserve:      # give the eager run the slab capacity the graph will bind (same split-N layout)
                        from spatten_amd import kv_slab
                        past = kv_slab.reserve(past, past[0][0].shape[2] + hz)
                    trace.append(tok.item())
                    for _ in range(gen - 1):
                        o = model(input_ids=tok, past_key_values=past, use_cache=True)
                        past, tok = o.past_key_values, o.logits[:, -1, :].argmax(-1).unsqueeze(1)
                        trace.append((tok.item(), o.logits[:, -1].clone()))
            return past, trace
        pa, ta = session(a, caches[0], reserve=bool(os.environ.get("FUZZ_ANY_HORIZON")))
        pb, tb = session(b, caches[1])
        for x, y in zip(ta, tb):
            if isinstance(x, tuple):
                assert x[0] == y[0] and torch.equal(x[1], y[1]), "token / logits"
            else:
                assert x == y, "first token"
        ext = getattr(caches[0], "ext", None)
        for i, ((ka, va), (kb, vb)) in enumerate(zip(pa, pb)):
            hk = slice(None) if ext is None or ext.layers[i].head_ids is None else ext.layers[i].head_ids.long()
            assert torch.equal(ka[:, hk], kb[:, hk]) and torch.equal(va[:, hk], vb[:, hk]), "cache"
        print("ok  ", tag, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", tag, "->", type(e).__name__, str(e)[:300], flush=True)
print(f"{n - bad} / {n} sessions agree")
