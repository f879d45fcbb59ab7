# session 3, call J: the final-build check — build() + smoke(), the full GPU suite, the bench lines of every config
cd $GRAFT_REPO_ROOT
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s3_full_suite.log 2>&1
grep -E "passed|failed|error" gpurun_out/s3_full_suite.log | tail -3
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > gpurun_out/r06_bench_default_run.json
timeout 900 python bench.py --config c3 --no-extras --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_c3.log | tail -1 > gpurun_out/r06_bench_c3.json
timeout 900 python bench.py --config c5 --no-extras --no-cpu-baseline > gpurun_out/bench_c5.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_c5.log | tail -1 > gpurun_out/r06_bench_c5.json
python - <<'PY'
import json
for f in ("default_run", "c3", "c5"):
    try:
        d = json.loads(open(f"gpurun_out/r06_bench_{f}.json").read())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"]["launch"][:40])
    except Exception as e:
        print(f, "ERR", e)
PY
