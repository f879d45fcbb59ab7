# Round-5 profile set (run on the GPU box through gpurun; the r05_* files land in gpurun_out/ — copy them into profiles/):
#   rocprofv3 kernel trace of the bench command (+ its JSON line), PMC traffic of the decode kernel, SQ counters of the prefill
#   flash kernel and of pqv_decode_kernel, kernel traces of the PQ profiles, the PQ-keyed prefill and one-launch local V.
# Every profiler run is bounded (timeout); PMC passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_bench $O/pmc_pf $O/pmc_pf2 $O/pmc_pf3 $O/pmc_pq $O/pmc_pq2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
grep "^{\"metric\"" $O/prof_bench.log | tail -1 > $O/r05_bench_line_under_rocprof.json
python $R/tools/trim_stats.py $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r05_bench_kernel_stats.csv
head -4 $O/r05_bench_kernel_stats.csv | cut -c1-200
timeout 300 bash $R/tools/pmc_decode.sh 2081 > $O/pmc_decode.log 2>&1
cp $O/pmc_decode.json $O/r05_pmc_decode.json
timeout 800 bash $R/tools/pmc_prefill.sh > $O/r05_pmc_prefill_raw.txt 2>&1
tail -30 $O/r05_pmc_prefill_raw.txt
# SQ counters of the profiled-plane decode kernel (VALU per byte): MSB pass of the three profiles at 8192 rows x 32 heads
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_pq -o p -- python $R/tools/mb/pqv_exp.py > /dev/null 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
fs = glob.glob(f"{O}/pmc_pq/**/*counter_collection.csv", recursive=True)
out = open(f"{O}/r05_pmc_pqv.txt", "w")
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "pqv_decode_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print(k, file=out)
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} launches {len(v):4d}  avg {sum(v) / len(v):16.0f}", file=out)
out.close()
print(open(f"{O}/r05_pmc_pqv.txt").read()[:3000])
PY
for probe in "pq_profiles tools/mb/pqv_exp.py" "local_v tools/mb/localv_exp.py" "prefill_pq tools/probe_prefill_pq.py"; do
  set -- $probe
  rm -rf $O/prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o p -- python $R/$2 > $O/r05_$1.txt 2>&1
  python $R/tools/trim_stats.py $(find $O/prof_$1 -name "*kernel_stats.csv" | head -1) $O/r05_$1_kernel_stats.csv
  grep -v amdgpu.ids $O/r05_$1.txt | tail -6
  head -5 $O/r05_$1_kernel_stats.csv | cut -c1-160
done
