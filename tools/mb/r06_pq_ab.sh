cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pq_profiles.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
python tools/mb/pqv_exp.py 32 8192 2>&1 | grep "V8\|V6\|bf16 K"
timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_c5.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["pq_confidence"], d["roofline"]["avg_launch_us"])
PY
