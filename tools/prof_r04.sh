# Round-4 profile set (run on the GPU box through gpurun; copy the r04_* files from gpurun_out/ into profiles/):
#   rocprofv3 kernel trace of the bench command, PMC traffic of the decode kernel, kernel traces of the round-4 kernels'
#   microbenchmarks (profiled quantised planes, one-launch local V pruning, fused projection + attention step).
# Every profiler run is bounded (timeout): a wedged run must not eat the GPU budget.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r04_bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
head -4 $O/r04_bench_kernel_stats.csv | cut -c1-200
timeout 300 bash $R/tools/pmc_decode.sh 2081 > $O/pmc_decode.log 2>&1
cp $O/pmc_decode.json $O/r04_pmc_decode.json
for probe in "pq_profiles tools/mb/pqv_exp.py" "local_v tools/mb/localv_exp.py" "fused_step tools/mb/fused_exp.py"; do
  set -- $probe
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o p -- python $R/$2 > $O/r04_$1.txt 2>&1
  python $R/tools/trim_stats.py $(find $O/prof_$1 -name "*kernel_stats.csv" | head -1) $O/r04_$1_kernel_stats.csv
  grep -v amdgpu.ids $O/r04_$1.txt | tail -6
  head -5 $O/r04_$1_kernel_stats.csv | cut -c1-160
done
# the paired-block prefill at q = N = 2048 (one shape per run: the file's average IS that shape's kernel time)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf2048 -o p -- python $R/tools/probe_prefill_shape.py 2048 2048 > $O/r04_prefill_q2048_n2048.txt 2>&1
python $R/tools/trim_stats.py $(find $O/prof_pf2048 -name "*kernel_stats.csv" | head -1) $O/r04_prefill_q2048_n2048_kernel_stats.csv
grep -v amdgpu.ids $O/r04_prefill_q2048_n2048.txt | tail -1
head -4 $O/r04_prefill_q2048_n2048_kernel_stats.csv | cut -c1-160
