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
    # round 6: `value` is the torch mirror on all pinned cores of one socket (median of three, spread printed), the one-thread
    # legs of the mirror AND of the C port beside it
    assert c["cores"] == c["pinned_cpus"] and len(c["repetitions"]) == 3 and c["spread"] >= 0
    assert c["one_thread"]["torch_mirror"]["value"] > 0 and ("c_port" in c["one_thread"] or "c_port_error" in c)
    # round 6: the default launch is the chained one (ONE launch per token); the per-layer launches are measured beside it
    assert "chained" in d["config"]["launch"] and r["layers_per_launch"] == d["config"]["layers"]
    assert abs(r["avg_launch_us"] - r["us_per_layer_step"] * r["layers_per_launch"]) < 0.05 * r["avg_launch_us"]
    pl = d["per_layer_launch"]
    assert pl["tokens_per_s"] > 0 and abs(pl["chained_over_per_layer"] - d["value"] / pl["tokens_per_s"]) < 1e-2
    assert d["scaling"] == "strong"


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_bench_single_gpu_lines_of_the_other_configs_carry_their_own_roofline(config):
    """`roofline.traffic` is the PMC figure of THIS config's kernel (profiles/*pmc_decode_<config>.json) or null — never another
    config's."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "66", "--warmup", "2",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    r = d["roofline"]
    assert d["config"]["name"] == config and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"], (r["traffic"], r["algorithmic_bytes_per_launch"])
    if r["traffic"] is not None:
        assert f"pmc_decode_{config}.json" in r["traffic_source"]
    assert ("chained" in d["config"]["launch"]) == (config == "c3")      # c5's quantised-plane steps keep one launch per layer
    if config == "c5":      # round 6: the default c5 line runs at the traces' refetch rate (~7 % of the heads), the worst case beside it
        pc = d["config"]["pq_confidence"]
        assert pc["confidence"] == "trace" and 0.03 <= pc["refetch_fraction"] <= 0.12, pc
        assert pc["uniform_refetch_fraction"] == 1.0 and pc["uniform_tokens_per_s"] < d["value"]
        assert pc["bf16_keys_same_steps_us_per_layer"] > 0


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
    assert d["scaling"] == "strong" and "one launch per layer" in c["launch"]     # (a collective between the layers: no chain)
    kept = c["heads_launched_per_layer_this_rank"]
    assert len(kept) == c["layers"] and all(k == (24 if config == "c3" else 30) for k in kept)
    if config == "c5":
        assert c["pq_profile"]["key_msb_bits"] == 8 and c["heads"] == 40 and c["kv_len_before_prune"] == 16384
    assert d["value"] > 0 and c["prune_events_in_timed_region"] == 2 and c["batch"] == (2 if config == "c3" else 1)
