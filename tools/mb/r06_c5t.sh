cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_pq_profiles.py tests/test_gpu_graph_decode.py tests/test_gpu_fullsize.py tests/test_gpu_hf_decoder.py tests/test_gpu_e2e_protocol.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
