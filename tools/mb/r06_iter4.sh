cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_fullsize.py tests/test_gpu_hf_decoder.py tests/test_gpu_e2e_protocol.py tests/test_gpu_cascade.py tests/test_gpu_bench_contract.py tests/test_gpu_integration_doc.py -x -q 2>&1 | tail -8
