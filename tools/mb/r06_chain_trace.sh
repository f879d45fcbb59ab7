cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so
timeout 300 python tools/mb/chain_trace.py 32 2081 32 > gpurun_out/chain_trace.log 2>&1
timeout 300 python tools/mb/chain_trace.py 4 2081 32 >> gpurun_out/chain_trace.log 2>&1
cat gpurun_out/chain_trace.log
