# session 3, call I: one tile per wave as the shortest chunk + the refitted default rule: both forms and the default at nine shapes
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gqa.py tests/test_gpu_gemv.py -x -q 2>&1 | tail -2
for shape in "32 8 1024" "32 8 2048" "32 8 3072" "32 8 4096" "32 8 8192" "64 8 1024" "64 8 2048" "16 8 8192" "32 4 4096" "32 16 4096" "28 4 4096" "32 8 4096 4"; do
  GQA_MODES=0,1,-1 timeout 200 python tools/mb/gqa_bench.py $shape 2>&1 | grep "mode=" | awk '{printf "%s %s %s %s %s %s %s | ", $1,$2,$3,$4,$7,$8,$9} END {print ""}'
done
