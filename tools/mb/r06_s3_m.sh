# session 3, call M: SQ counters of the flash kernel on the round-6 build, reference and fp32-logit numerics
cd $GRAFT_REPO_ROOT
timeout 120 python tools/probe_prefill.py 2>&1 | grep -v amdgpu.ids | tail -8
bash tools/pmc_prefill.sh > gpurun_out/r06_pmc_prefill_raw.txt 2>&1
cat gpurun_out/r06_pmc_prefill_raw.txt | cut -c1-200 | tail -70
