cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for i in 1 2; do for l in /tmp/lib_keep.so tools/mb/ab/lib_pqvd3.so tools/mb/ab/lib_pqvd3u2.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/mb/pqv_exp.py 32 8192 2>&1 | grep -E "V8|V6"; python tools/mb/pqv_exp.py 40 8192 2>&1 | grep -E "8\+4, V8"; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python $R/bench.py --config c5 --steps 64 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_c5.log 2>&1
grep "^{\"metric\"" $O/prof_c5.log | tail -1 | cut -c1-600
python $R/tools/trim_stats.py $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/r05_c5_kernel_stats.csv
head -12 $O/r05_c5_kernel_stats.csv | cut -c1-220
