# Round 6: the bench lines quoted in README / DESIGN (driver-style, default, c3, c5) -> gpurun_out/r06_bench_*.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style.json 2> gpurun_out/r06_bench_driver_style.err
timeout 1200 python bench.py --no-extras > gpurun_out/r06_bench_default_run.json 2> gpurun_out/r06_bench_default_run.err
timeout 900 python bench.py --config c3 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_c3.json 2> gpurun_out/r06_bench_c3.err
timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err
python - <<'PY'
import json
for f in ("r06_bench_driver_style", "r06_bench_default_run", "r06_bench_c3", "r06_bench_c5"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("per_layer_launch"), {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "us_per_layer_step", "traffic", "algorithmic_bytes_per_launch")}, d.get("prune_event", {}).get("us_all_layers"), d.get("prune_event", {}).get("gather_only_frac_of_hbm_peak"), d["config"].get("pq_confidence"))
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]; print("  cpu", c["value"], c["cores"], c["spread"], c.get("best_of_thread_counts"), c["one_thread"]["torch_mirror"]["value"], c["one_thread"].get("c_port", {}).get("value"), c.get("kept_set_vs_reference_c2"))
        if "extras" in d:
            e = d["extras"]; print("  extras", {k: e[k] for k in e if "speedup" in k or "prefill_8192_causal" in k or "prefill_2048" in k or k in ("per_rank_launch_us", "per_rank_chained_us_per_layer", "plugin_path_graph_tokens_per_s", "dense_fused_tokens_per_s", "c5_local_v_30pct_vs_plain_decode")})
    except Exception as e:
        print(f, "ERR", e)
PY
