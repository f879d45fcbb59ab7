# session 3, call E: the merging workgroup's first completion-word poll in front of its tile requests (A/B, stamps);
# where the grouped-query matrix-core form starts to win (group size x rows)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for lib in spatten_amd/lib/libspatten_hip.so tools/mb/ab/lib_chain_pf.so; do
    echo "== $lib"
    for h in 32 24 4; do
      SPATTEN_LIB=$PWD/$lib timeout 300 python tools/mb/chain_bench.py $h 2081 32 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chain_pf.so timeout 600 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
SPATTEN_LIB=$PWD/tools/mb/ab/lib_chaintrace_pf.so timeout 300 python tools/mb/chain_trace.py 32 2081 32 2>&1 | grep -v amdgpu.ids | tail -13
for shape in "32 8 2048" "32 8 4096" "32 8 6144" "32 8 8192" "64 8 1024" "64 8 2048" "64 8 4096" "16 8 4096" "16 8 8192" "16 8 16384" "32 4 8192" "32 16 8192"; do
  GQA_MODES=0,1 timeout 200 python tools/mb/gqa_bench.py $shape 2>&1 | grep "mode=" | tail -2
done
