# Round-6 profile set of the grouped-query decode step (32 query heads on 8 kv heads x 16384 rows, tools/mb/gqa_bench.py, matrix-core form):
# rocprofv3 kernel trace, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only; FETCH doubled per the guide's
# gfx950 correction, unit KiB), phase stamps (needs tools/mb/ab/lib_gqatrace.so).  Files land in gpurun_out/ — copy into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export GQA_MODES=1,1
rm -rf $O/prof_gqa
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gqa -o g -- python $R/tools/mb/gqa_bench.py 32 8 16384 > $O/prof_gqa.log 2>&1
python $R/tools/trim_stats.py $(find $O/prof_gqa -name "*kernel_stats.csv" | head -1) $O/r06_gqa_kernel_stats.csv
head -3 $O/r06_gqa_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_gqa_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_gqa_$c -o p -- python $R/tools/mb/gqa_bench.py 32 8 16384 > $O/pmc_gqa_$c.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
out = {"workload": "tools/mb/gqa_bench.py 32 8 16384 (bf16, d = 128, B = 1; matrix-core form, append + stash inside the launch)", "kernel": "decode_gqa_kernel"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/pmc_gqa_{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "decode_gqa_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KiB_per_launch"] = sum(v) / max(1, len(v)); out[c + "_launches"] = len(v)
alg = 2 * 8 * 16384 * 128 * 2 + 2 * 32 * 128 * 2 + 32 * 16384 * 2
out["algorithmic_bytes_per_launch"] = alg
out["traffic_bytes_per_launch"] = (2 * out["FETCH_SIZE_KiB_per_launch"] + out["WRITE_SIZE_KiB_per_launch"]) * 1024
out["traffic_over_algorithmic"] = round(out["traffic_bytes_per_launch"] / alg, 3)
json.dump(out, open(f"{O}/r06_pmc_gqa.json", "w"), indent=1)
print(json.dumps(out))
PY
unset GQA_MODES
SPATTEN_LIB=$R/tools/mb/ab/lib_gqatrace.so timeout 300 python $R/tools/mb/gqa_trace.py 32 8 16384 2>&1 | grep -v amdgpu.ids > $O/r06_gqa_phase_stamps.txt
tail -13 $O/r06_gqa_phase_stamps.txt
