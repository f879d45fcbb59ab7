cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -5
: > gpurun_out/chain_merger.log
timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=7 timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=7 CHAIN_LAYOUT=2240 timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=6 timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=8 timeout 300 python tools/mb/chain_bench.py 24 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=9 timeout 300 python tools/mb/chain_bench.py 24 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
CHAIN_MERGER=1 CHAIN_SPLITS=8 timeout 300 python tools/mb/chain_bench.py 4 2081 32 2>&1 | tail -1 >> gpurun_out/chain_merger.log
cat gpurun_out/chain_merger.log
