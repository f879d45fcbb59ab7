# session 3, call N: the grouped-query step with two 16-byte stash stores per tile (was four 8-byte ones)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gqa.py -x -q 2>&1 | tail -2
for shape in "32 8 16384" "32 8 8192" "32 8 4096" "64 8 8192"; do
  GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py $shape 2>&1 | grep "mode=" | tail -1
done
