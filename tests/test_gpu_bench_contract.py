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


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_bench_runs_the_8_gpu_configs_of_baseline_json_on_one_rank_through_rccl(config):
    """BASELINE.json configs[2] (C3) and configs[4] (C5) are head-parallel configurations: `bench.py --config c3|c5` must be
    valid at any --gpus; here one rank with --force-dist (the RCCL communicator, the per-layer exchange inside the graph)."""
    # (c3 with a batch of 2: the per-rank shape of a weak-scaling run at 2 GPUs — B sequences x the rank's heads)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", "1", "--steps", "66",
                          "--warmup", "2", "--force-dist"] + (["--batch", "2"] if config == "c3" else []),
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    c = d["config"]
    assert c["name"] == config and "configs[" in c["workload"]
    assert c["rccl_ranks"] == 1 and c["exchange"] == "per-layer" and d["comm"]["comm_us_per_token"] is not None
    kept = c["heads_launched_per_layer_this_rank"]
    assert len(kept) == c["layers"] and all(k == (24 if config == "c3" else 30) for k in kept)
    if config == "c5":
        assert c["pq_profile"]["key_msb_bits"] == 8 and c["heads"] == 40 and c["kv_len_before_prune"] == 16384
    assert d["value"] > 0 and c["prune_events_in_timed_region"] == 2 and c["batch"] == (2 if config == "c3" else 1)
