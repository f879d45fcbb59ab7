# Round-3 profile set (run on the GPU box through gpurun; copy the r03_* files from gpurun_out/ into profiles/):
#   rocprofv3 kernel trace of the bench command, PMC traffic of the decode kernel, ONE kernel-stats file PER prefill shape,
#   kernel trace of the drop-in decode step under its captured graph, decode anatomy A/B (stripped kernels).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r03_bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
head -4 $O/r03_bench_kernel_stats.csv | cut -c1-200
bash $R/tools/pmc_decode.sh 2081 > $O/pmc_decode.log 2>&1
cp $O/pmc_decode.json $O/r03_pmc_decode.json
# per-shape prefill kernel stats: one rocprofv3 run per shape so that each file's average IS that shape's kernel time
for shape in "2048 2048" "8192 8192" "8192 8192 fast" "64 2112"; do
  set -- $shape
  tag="q$1_n$2${3:+_$3}"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf_$tag -o pf -- python $R/tools/probe_prefill_shape.py $shape > $O/r03_prefill_${tag}.txt 2>&1
  python $R/tools/trim_stats.py $(find $O/prof_pf_$tag -name "*kernel_stats.csv" | head -1) $O/r03_prefill_${tag}_kernel_stats.csv
  tail -1 $O/r03_prefill_${tag}.txt
done
# the drop-in decode step as one captured graph (fused q/k/v + native projections): what a token is made of
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pg -o pg -- python $R/tools/probe_plugin_graph.py 111 > $O/r03_plugin_graph_probe.txt 2>&1
python $R/tools/trim_stats.py $(find $O/prof_pg -name "*kernel_stats.csv" | head -1) $O/r03_plugin_graph_kernel_stats.csv
head -6 $O/r03_plugin_graph_kernel_stats.csv | cut -c1-200
# decode anatomy: the shipped kernel, without the split merge, without the in-workgroup reduction (wrong results: timing only)
cd $R && bash tools/mb/dec_exp.sh "" "-DSPATTEN_EXP_NOMERGE" "-DSPATTEN_EXP_NOREDUCE" > $O/r03_decode_anatomy_ab.txt 2>&1
cat $O/r03_decode_anatomy_ab.txt
