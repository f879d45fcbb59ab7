cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_mb -o t -- $R/tools/mb/pattern > /dev/null 2>&1
find $R/gpurun_out/p_mb -name "*kernel_stats.csv" | head -1 | xargs cat | head -6
