# rocprofv3 kernel trace of the prefill probe (q = N = 2048 / 8192 causal, +column importance, +stash at 4096)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prefill -o pf -- python $R/tools/probe_prefill.py > $R/gpurun_out/prof_prefill.log 2>&1
python $R/tools/trim_stats.py $(find $R/gpurun_out/prof_prefill -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prefill_kernel_stats.csv
grep "prefill N" $R/gpurun_out/prof_prefill.log
head -8 $R/gpurun_out/prefill_kernel_stats.csv | cut -c1-180
