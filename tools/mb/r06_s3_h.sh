# session 3, call H: the shortest chunk a split of the grouped-query step may get (256 rows = two tiles per wave shipped): short caches
cd $GRAFT_REPO_ROOT
for mc in 256 128 64; do
  for shape in "32 8 2048" "32 8 4096" "32 8 8192" "64 8 2048" "64 8 4096"; do
    SPATTEN_GQA_MIN_CHUNK=$mc GQA_MODES=1,1 timeout 200 python tools/mb/gqa_bench.py $shape 2>&1 | grep "mode=" | tail -1 | sed "s/^/min_chunk=$mc /"
  done
done
