cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_fullsize.py tests/test_gpu_cascade.py tests/test_gpu_hf_decoder.py tests/test_gpu_e2e_protocol.py tests/test_gpu_random_sweep.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
python tools/mb/pf8k_probe.py 2>&1 | grep prefill
