# Round-2 profile set: rocprofv3 kernel trace of the bench command, PMC traffic of the decode kernel, kernel trace of
# the prefill probe.  Writes under gpurun_out/ (copy into profiles/ as r02_*).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_bench.log 2>&1
grep "^{\"metric\"" $R/gpurun_out/prof_bench.log | tail -1 > $R/gpurun_out/r02_bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_bench_kernel_stats.csv
head -8 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-200
bash $R/tools/pmc_decode.sh 2081 > $R/gpurun_out/pmc_decode.log 2>&1
cp $R/gpurun_out/pmc_decode.json $R/gpurun_out/r02_pmc_decode.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pf -o pf -- python $R/tools/probe_prefill.py > $R/gpurun_out/r02_prefill_probe_under_rocprof.txt 2>&1
python $R/tools/trim_stats.py $(find $R/gpurun_out/prof_pf -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_prefill_kernel_stats.csv
