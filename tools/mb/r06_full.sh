cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/gpu_tests.log
cat gpurun_out/gpu_tests.log
