"""The one-wave-per-SIMD flash kernel (tools/experiments/prefill_w4.h — an experiment, selected by SPATTEN_PREFILL_W4=1)
against the oracle.  The kernel choice is read once per process, so the cases run in a subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CASES = r'''
import ctypes, sys
import numpy as np
import torch
from oracle import spatten_oracle as orc
from spatten_amd import _lib
from tests.util import OUT_TOL, attn_inputs
from tests.test_gpu_prefill import run_prefill

lib = _lib.load()
n = 0
for dt in ("bf16", "f16"):
    for (B, H, Hkv, P, ql, causal) in [(2, 4, 2, 0, 130, True), (2, 4, 2, 300, 200, True), (1, 4, 4, 513, 64, True),
                                       (1, 8, 8, 0, 700, True), (1, 4, 2, 100, 333, True), (1, 4, 4, 37, 256, True),
                                       (1, 4, 4, 0, 257, False), (1, 2, 2, 64, 1100, True)]:
        d = 128
        q, k, v, past = attn_inputs(B, H, Hkv, d, P, ql, dt, seed=900 + P + ql)
        N = P + ql
        pos = np.tile(np.arange(P, N)[None], (B, 1))
        mask = orc.causal_mask(B, ql, N, dt) if causal else None
        o, _, _ = orc.attention_core(q, k, v, None if past is None else past[0], None if past is None else past[1], pos, mask, dt)
        out, _, _ = run_prefill(q, k, v, past, dt, causal=causal, stash=False)
        assert lib.spatten_debug_last_prefill_kernel() == 2, "the w4 kernel did not run"
        np.testing.assert_allclose(out, o, err_msg=f"{dt} {B} {H} {Hkv} {P} {ql} {causal}", **OUT_TOL[dt])
        n += 1
# fast numerics: unit-variance logits, the stated tolerance of test_prefill_fast_numerics_stays_within_the_stated_tolerance
q, k, v, past = attn_inputs(1, 4, 4, 128, 0, 600, "bf16", seed=77)
pos = np.arange(600)[None]
o, _, _ = orc.attention_core(q, k, v, None, None, pos, orc.causal_mask(1, 600, 600, "bf16"), "bf16")
out, _, _ = run_prefill(q, k, v, None, "bf16", causal=True, stash=False, numerics="fast")
assert lib.spatten_debug_last_prefill_kernel() == 2
np.testing.assert_allclose(out, o, **OUT_TOL["bf16"])
print("W4_CASES_OK", n + 1)
'''


def test_w4_kernel_matches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, SPATTEN_PREFILL_W4="1", SPATTEN_PREFILL_VTR="0", SPATTEN_PREFILL_KSPLIT="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", CASES], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "W4_CASES_OK 17" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
