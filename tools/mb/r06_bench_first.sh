cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err
tail -c 3000 gpurun_out/bench_driver_style.err
timeout 900 python bench.py --config c3 --no-extras --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
python - <<'PY'
import json
for f in ("bench_driver_style", "bench_c3", "bench_c5"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["config"]["launch"][:40], d.get("per_layer_launch"), {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "us_per_layer_step", "traffic")})
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]; print("  cpu", c["value"], c["cores"], c["spread"], c.get("best_of_thread_counts"), c["one_thread"]["torch_mirror"]["value"], c["one_thread"].get("c_port", {}).get("value"))
        if "extras" in d:
            e = d["extras"]; print("  extras", {k: e[k] for k in e if "speedup" in k or "prefill_8192_causal_TF" in k or k == "per_rank_launch_us"})
    except Exception as e:
        print(f, "ERR", e)
PY
