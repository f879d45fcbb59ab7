# the bench lines of the final build (default, driver-style, c3, c5) and the round-5 profile set (tools/prof_r05.sh); outputs in gpurun_out/
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r05_bench_default_run.json 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_driver_style.json 2>/dev/null
python bench.py --config c3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' > gpurun_out/r05_bench_c3.json
python bench.py --config c5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' > gpurun_out/r05_bench_c5.json
bash tools/prof_r05.sh > gpurun_out/prof_r05.log 2>&1
tail -3 gpurun_out/prof_r05.log
