"""Randomised geometry for the end-to-end protocol tests (developer tool): tests/test_gpu_e2e_protocol.py's GPU-vs-oracle-replica
comparisons re-run with other layer / head / head_dim / window sizes.   python tools/fuzz_e2e.py [n] [seed]"""
import io
import os
import random
import sys
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_e2e_protocol as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    E.L, E.H, E.D = rng.choice([2, 3]), rng.choice([2, 4, 8]), rng.choice([64, 128])
    E.HID = E.H * E.D
    E.IMPORTANT, E.RECENT, E.MAX_GEN = rng.randint(12, 22), rng.randint(16, 24), rng.randint(6, 9)
    tag = f"L={E.L} H={E.H} D={E.D} imp={E.IMPORTANT} rec={E.RECENT} gen={E.MAX_GEN}"
    for name, fn in [("cascade", E.test_multi_turn_protocol_cascade_importance_mode)] + \
                    [(m, (lambda m=m: E.test_multi_turn_protocol_extension_modes(m))) for m in ("head", "pq", "local_v", "head+pq+cascade")] + \
                    [("layer_cascade", E.test_multi_turn_protocol_layer_cascade)]:
        if (name in ("head", "head+pq+cascade") and E.H < 4) or (name == "layer_cascade" and E.L != 2):
            continue            # (the head modes keep 3 heads; the layer-cascade test spells out two per-layer keeps)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fn()
            print("ok  ", tag, name, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, name, "->", type(e).__name__, str(e)[:200].replace("\n", " "), flush=True)
print("failures:", bad)
