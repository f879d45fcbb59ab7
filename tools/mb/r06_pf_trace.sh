cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_pf8k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf8k -o p -- python $R/tools/mb/pf8k_probe.py > $O/r06_prefill_q8192_n8192.txt 2>&1
python $R/tools/trim_stats.py $(find $O/prof_pf8k -name "*kernel_stats.csv" | head -1) $O/r06_prefill_q8192_n8192_kernel_stats.csv
grep prefill $O/r06_prefill_q8192_n8192.txt; head -4 $O/r06_prefill_q8192_n8192_kernel_stats.csv | cut -c1-200
