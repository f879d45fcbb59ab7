cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pq_profiles.py tests/test_gpu_graph_decode.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 > gpurun_out/c5_tests.log
for c in 128 256 512 1024; do
SPATTEN_PQV_COMPACT=$c timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/bench_c5_$c.json 2> gpurun_out/bench_c5_$c.err
python - <<PY >> gpurun_out/c5_tests.log
import json
d = json.loads([l for l in open("gpurun_out/bench_c5_$c.json") if l.startswith("{")][-1])
print("compact=$c", d["value"], d["ms_per_step"], d["config"]["pq_confidence"]["uniform_tokens_per_s"], d["roofline"]["avg_launch_us"])
PY
done
cat gpurun_out/c5_tests.log
