cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r04_bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
head -4 $O/r04_bench_kernel_stats.csv | cut -c1-220
timeout 300 bash $R/tools/pmc_decode.sh 2081 > $O/pmc_decode.log 2>&1
cp $O/pmc_decode.json $O/r04_pmc_decode.json; cat $O/r04_pmc_decode.json | head -14
