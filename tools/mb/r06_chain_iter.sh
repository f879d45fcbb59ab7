cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -15 > gpurun_out/chain_test.log
timeout 300 python tools/mb/chain_bench.py > gpurun_out/chain_bench.log 2>&1
timeout 300 python tools/mb/chain_bench.py 24 2081 32 >> gpurun_out/chain_bench.log 2>&1
timeout 300 python tools/mb/chain_bench.py 4 2081 32 >> gpurun_out/chain_bench.log 2>&1
timeout 300 python tools/mb/chain_bench.py 40 8192 8 >> gpurun_out/chain_bench.log 2>&1
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 > gpurun_out/chain_trace.log 2>&1
cat gpurun_out/chain_test.log gpurun_out/chain_bench.log gpurun_out/chain_trace.log
