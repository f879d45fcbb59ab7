# Round-6 profile set (run on the GPU box through gpurun; the r06_* files land in gpurun_out/ — copy them into profiles/):
#   rocprofv3 kernel trace of the bench command (+ its JSON line) for c2 (chained default AND --launch per-layer), c3, c5;
#   PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of each config's dominant kernel;
#   kernel trace of the 8192-token prefill launch.
# Every profiler run is bounded (timeout); PMC passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for cfg in "c2 chained" "c2 per-layer" "c3 chained" "c5 auto"; do
  set -- $cfg
  tag=$1_$(echo $2 | tr - _)
  rm -rf $O/prof_bench_$tag
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_$tag -o bench -- python $R/bench.py --config $1 --launch $2 --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_bench_$tag.log 2>&1
  grep "^{\"metric\"" $O/prof_bench_$tag.log | tail -1 > $O/r06_bench_${tag}_line_under_rocprof.json
  python $R/tools/trim_stats.py $(find $O/prof_bench_$tag -name "*kernel_stats.csv" | head -1) $O/r06_bench_${tag}_kernel_stats.csv
  head -4 $O/r06_bench_${tag}_kernel_stats.csv | cut -c1-220
done
cp $O/r06_bench_c2_chained_kernel_stats.csv $O/r06_bench_kernel_stats.csv
for cfg in c2 c3 c5; do
  timeout 600 bash $R/tools/pmc_bench.sh $cfg > $O/pmc_bench_$cfg.log 2>&1
  cp $O/pmc_decode_$cfg.json $O/r06_pmc_decode_$cfg.json
  tail -1 $O/pmc_bench_$cfg.log | cut -c1-600
done
rm -rf $O/prof_pf8k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf8k -o p -- python $R/tools/mb/pf8k_probe.py > $O/r06_prefill_q8192_n8192.txt 2>&1
python $R/tools/trim_stats.py $(find $O/prof_pf8k -name "*kernel_stats.csv" | head -1) $O/r06_prefill_q8192_n8192_kernel_stats.csv
grep -v amdgpu.ids $O/r06_prefill_q8192_n8192.txt | tail -5
head -5 $O/r06_prefill_q8192_n8192_kernel_stats.csv | cut -c1-200
