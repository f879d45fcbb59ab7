cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_local_v.py tests/test_gpu_pq_profiles.py tests/test_gpu_decode.py tests/test_gpu_chain.py -x -q 2>&1 | tail -8
timeout 900 bash tools/pmc_c5.sh 2>&1 | tail -2 | cut -c1-900
cp gpurun_out/pmc_decode_c5.json gpurun_out/r06_pmc_decode_c5.json
