"""The binding stub printed in INTEGRATION.md §2 is executed as written (only the library path is substituted) and its
``prune_one_layer`` is checked against the oracle: the document cannot drift from the ABI."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import spatten_oracle as orc
from tests.util import dev, host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_binding_stub_runs_and_matches_the_oracle():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# spatten_llm/_hip\.py.*?)```", text, re.S).group(1)
    so = os.path.join(ROOT, "spatten_amd", "lib", "libspatten_hip.so")
    ns = {}
    exec(compile(block.replace("/path/to/libspatten_hip.so", so), "INTEGRATION.md", "exec"), ns)
    dt, B, H, L, d = "bf16", 1, 6, 700, 64
    start, recent, important, coming = 4, 100, 200, 30
    K = orc.synth_normal(5, 0, (B, H, L, d), dt)
    V = orc.synth_normal(5, 1, (B, H, L, d), dt)
    stash = orc.synth_normal(5, 2, (B, H, 1, L), dt)
    want, idx = orc.apply_token_pruning([(K, V)], coming, [stash], start, recent, important, dt)
    score = dev(stash[0, :, 0, :], dt)                     # importance of a single-token stash = the stash row (:51)
    hi = L - recent + coming
    Kn, Vn = ns["prune_one_layer"](score, dev(K, dt), dev(V, dt), start, hi, important)
    torch.cuda.synchronize()
    assert np.array_equal(host(Kn), want[0][0]) and np.array_equal(host(Vn), want[0][1])
