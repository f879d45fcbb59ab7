cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -8
python bench.py --config c5 --steps 64 --warmup 64 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | cut -c1-330
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python $R/bench.py --config c5 --steps 64 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_c5.log 2>&1
python $R/tools/trim_stats.py $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/r05_c5_kernel_stats.csv
head -6 $O/r05_c5_kernel_stats.csv | cut -c1-200
