# session 3, call G: profile set of the grouped-query step, the chain's finer stamps on the final build, the driver-style bench line
cd $GRAFT_REPO_ROOT
bash tools/prof_r06_gqa.sh
cd $GRAFT_REPO_ROOT
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_phase_stamps.txt
tail -14 gpurun_out/r06_chain_phase_stamps.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_driver_style.log | tail -1 > gpurun_out/r06_bench_driver_style.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_driver_style.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["per_layer_launch"]["tokens_per_s"])
print({k: v for k, v in d["extras"].items() if k.startswith("gqa") or "error" in k})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
