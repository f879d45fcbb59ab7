cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/chain_splits.log
for cfg in "24 8" "24 10" "16 8" "16 12" "16 16" "8 8" "8 16" "8 32" "4 8" "4 16" "4 32" "4 64"; do
  set -- $cfg
  CHAIN_SPLITS=$2 timeout 300 python tools/mb/chain_bench.py $1 2081 32 2>&1 | tail -1 >> gpurun_out/chain_splits.log
done
cat gpurun_out/chain_splits.log
