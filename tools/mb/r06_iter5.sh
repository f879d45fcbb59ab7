cd $GRAFT_REPO_ROOT
timeout 600 bash tools/pmc_c5.sh trace 2>&1 | tail -1 | cut -c1-700
cp gpurun_out/pmc_decode_c5_trace.json gpurun_out/r06_pmc_decode_c5.json
timeout 600 bash tools/pmc_c5.sh uniform 2>&1 | tail -1 | cut -c1-300
cp gpurun_out/pmc_decode_c5_uniform.json gpurun_out/r06_pmc_decode_c5_uniform_refetch_all.json
