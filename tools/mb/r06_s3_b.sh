# session 3, call B: the fixed dispatch (gemv + gqa tests), then what paces the grouped-query stream: split count / chunk length
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gqa.py tests/test_gpu_gemv.py -x -q 2>&1 | tail -3
for ns in 0 31 29 24 16 48 64; do
  GQA_NS=$ns GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 16384 2>&1 | grep "mode=" | tail -1
done
for n in 16000 17000 12288 8192; do
  GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py 32 8 $n 2>&1 | grep "mode=" | tail -1
done
# the same decomposition as plain multi-head steps (8 heads, register tiles): what the stream does without the group
GQA_MODES=0,0 timeout 200 python tools/mb/gqa_bench.py 8 8 16384 2>&1 | grep "mode=" | tail -1
GQA_MODES=0,0 timeout 200 python tools/mb/gqa_bench.py 8 8 4096 2>&1 | grep "mode=" | tail -1
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_gqa.py 2>&1 | tail -4
