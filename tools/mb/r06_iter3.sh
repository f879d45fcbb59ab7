cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pq_profiles.py -x -q -k "trace_like" 2>&1 | tail -5
timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; tail -3 gpurun_out/bench_c5.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_c5.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["pq_confidence"], {k: d["roofline"][k] for k in ("frac", "avg_launch_us", "traffic", "algorithmic_bytes_per_launch")})
PY
