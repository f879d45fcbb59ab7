"""bench.py's output contract: ONE JSON line (the last line of stdout) with the driver's keys, the roofline and
cpu_baseline objects.  Needs an MI355X (the bench has no CPU path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "70", "--warmup", "3",
                          "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 70 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3            # tokens/s = 1 / (s per token)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.05 < r["frac"] < 1.0 and (r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
