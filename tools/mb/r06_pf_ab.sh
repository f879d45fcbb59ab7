cd $GRAFT_REPO_ROOT
SPATTEN_LIB=$PWD/tools/mb/ab/lib_pf_dma3.so timeout 1200 python -m pytest tests/test_gpu_prefill.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for l in "" tools/mb/ab/lib_pf_dma2.so tools/mb/ab/lib_pf_dma3.so "" tools/mb/ab/lib_pf_dma2.so tools/mb/ab/lib_pf_dma3.so; do
  echo "== lib ${l:-default}"
  SPATTEN_LIB=${l:+$PWD/$l} python tools/mb/pf8k_probe.py 2>&1 | grep prefill
done
