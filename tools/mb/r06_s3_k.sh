# session 3, call K: per-tile stamps of the grouped-query step (DMA rings), the register-tile fill (compiler-counted waits) beside it
cd $GRAFT_REPO_ROOT
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqatrace.so timeout 300 python tools/mb/gqa_trace.py 32 8 16384 2>&1 | grep -v amdgpu.ids | tail -19
SPATTEN_GQA_REGT=1 timeout 600 python -m pytest tests/test_gpu_gqa.py -x -q 2>&1 | tail -2
SPATTEN_GQA_REGT=0 GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 16384 2>&1 | grep "mode=" | tail -1
SPATTEN_GQA_REGT=1 GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 16384 2>&1 | grep "mode=" | tail -1
SPATTEN_GQA_REGT=1 GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 4096 2>&1 | grep "mode=" | tail -1
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqatrace_regt.so timeout 300 python tools/mb/gqa_trace.py 32 8 16384 2>&1 | grep -v amdgpu.ids | tail -19
