cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 4 3 1 0; do
  SPATTEN_DEBUG=$dbg SPATTEN_DECODE_UNR=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_$dbg -o t -- $R/tools/mb/decode_bench 1 2048 16 1 > /dev/null 2>&1
  echo "debug=$dbg"; find $R/gpurun_out/p_$dbg -name "*kernel_stats.csv" | head -1 | xargs cat | head -3
done
