# session 3, call O: the query entry point against the dispatch; the decode step's stash stores behind P.V instead of inside the scores (A/B, twice)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gqa.py tests/test_abi_exports.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for lib in spatten_amd/lib/libspatten_hip.so tools/mb/ab/lib_stashlate.so; do
    echo "== $lib"
    for h in 32 24; do
      SPATTEN_LIB=$PWD/$lib timeout 300 python tools/mb/chain_bench.py $h 2081 32 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-215
    done
  done
done
SPATTEN_LIB=$PWD/tools/mb/ab/lib_stashlate.so timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_decode.py -x -q 2>&1 | tail -2
