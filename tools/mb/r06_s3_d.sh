# session 3, call D: early merge loads in the chained launch (A/B against the build without, twice), its stamps, bit-identity tests;
# the grouped-query refill variants (nt / split refill)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  for lib in spatten_amd/lib/libspatten_hip.so tools/mb/ab/lib_chain_noearly.so; do
    echo "== $lib"
    SPATTEN_LIB=$PWD/$lib timeout 300 python tools/mb/chain_bench.py 32 2081 32 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
SPATTEN_LIB=$PWD/spatten_amd/lib/libspatten_hip.so timeout 300 python tools/mb/chain_bench.py 24 2081 32 2>&1 | grep -v amdgpu.ids | tail -2
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chain_noearly.so timeout 300 python tools/mb/chain_bench.py 24 2081 32 2>&1 | grep -v amdgpu.ids | tail -2
SPATTEN_LIB=$PWD/spatten_amd/lib/libspatten_hip.so timeout 300 python tools/mb/chain_bench.py 4 2081 32 2>&1 | grep -v amdgpu.ids | tail -2
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chain_noearly.so timeout 300 python tools/mb/chain_bench.py 4 2081 32 2>&1 | grep -v amdgpu.ids | tail -2
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 2>&1 | grep -v amdgpu.ids | tail -17
for v in 00 10 01 11 00 11; do
  echo "== gqa variant nt/split = $v"
  SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqa$v.so GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 16384 2>&1 | grep "mode=" | tail -1
done
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqa11.so GQA_MODES=1 timeout 200 python tools/mb/gqa_bench.py 32 8 4096 2>&1 | grep "mode=" | tail -1
SPATTEN_LIB=$PWD/tools/mb/ab/lib_gqa11.so GQA_MODES=1 timeout 200 python tools/mb/gqa_bench.py 64 8 8192 2>&1 | grep "mode=" | tail -1
timeout 600 python -m pytest tests/test_gpu_gqa.py -x -q 2>&1 | tail -2
