# session 3, call A: GQA state on this box (tests, bench, stamps) + the full GPU suite on the fresh build
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_gqa.py -x -q 2>&1 | tail -3
timeout 300 python tools/mb/gqa_bench.py 32 8 16384 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/mb/gqa_bench.py 32 8 4096 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/mb/gqa_bench.py 64 8 8192 2>&1 | grep -v amdgpu.ids | tail -4
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqatrace.so timeout 300 python tools/mb/gqa_trace.py 32 8 16384 2>&1 | grep -v amdgpu.ids | tail -14
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
