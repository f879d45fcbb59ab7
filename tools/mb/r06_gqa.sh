cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gqa.py -x -q 2>&1 | tail -5
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqatrace.so timeout 300 python tools/mb/gqa_trace.py 32 8 16384 2>&1 | tail -14
