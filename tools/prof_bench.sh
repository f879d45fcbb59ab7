# rocprofv3 kernel trace of the bench command; trimmed stats summary -> gpurun_out/ (copy into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_bench.log 2>&1
grep "^{\"metric\"" $R/gpurun_out/prof_bench.log | tail -1 > $R/gpurun_out/bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/bench_kernel_stats.csv
head -8 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-200
