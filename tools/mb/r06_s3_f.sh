# session 3, call F: the polling merger folds its own partial from registers (A/B, stamps); the grouped-query default rule
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gqa.py tests/test_gpu_gemv.py tests/test_gpu_chain.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for lib in spatten_amd/lib/libspatten_hip.so tools/mb/ab/lib_own.so; do
    echo "== $lib"
    for h in 32 24 4; do
      SPATTEN_LIB=$PWD/$lib timeout 300 python tools/mb/chain_bench.py $h 2081 32 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
SPATTEN_LIB=$PWD/tools/mb/ab/lib_own.so timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_decode.py tests/test_gpu_graph_decode.py -x -q 2>&1 | tail -2
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 2>&1 | grep -v amdgpu.ids | tail -13
GQA_MODES=-1 timeout 200 python tools/mb/gqa_bench.py 32 8 4096 2>&1 | grep "mode=" | tail -1
GQA_MODES=-1 timeout 200 python tools/mb/gqa_bench.py 32 8 8192 2>&1 | grep "mode=" | tail -1
