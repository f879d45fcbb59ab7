# session 3, call C: the full GPU suite (summary kept) + the chained launch's finer stamps (restart gap between two layer-steps)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s3_full_suite.log 2>&1
grep -E "passed|failed|error" gpurun_out/s3_full_suite.log | tail -3
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 2>&1 | grep -v amdgpu.ids | tail -34
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 4 2081 32 2>&1 | grep -v amdgpu.ids | tail -16
