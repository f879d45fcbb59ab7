cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/chain_layout.log
for lay in 0 2176 2240 2304 2400 2520; do
  CHAIN_LAYOUT=$lay timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | tail -1 >> gpurun_out/chain_layout.log
done
for lay in 0 2240 2400; do
  CHAIN_LAYOUT=$lay timeout 300 python tools/mb/chain_bench.py 24 2081 32 2>&1 | tail -1 >> gpurun_out/chain_layout.log
  CHAIN_LAYOUT=$lay timeout 300 python tools/mb/chain_bench.py 4 2081 32 2>&1 | tail -1 >> gpurun_out/chain_layout.log
done
cat gpurun_out/chain_layout.log
