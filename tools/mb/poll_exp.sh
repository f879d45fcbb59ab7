# A/B: single-hop polling merge vs ticket merge, same box, same build
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  SPATTEN_DECODE_POLL=$v python bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('poll=$v', d['value'], d['roofline']['avg_launch_us'])"
done
python -m pytest tests/test_gpu_decode.py tests/test_gpu_cascade.py -q 2>&1 | tail -3
