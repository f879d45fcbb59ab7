# HBM traffic of the decode kernel from PMC counters (separate rocprofv3 --pmc passes, kernel-trace only),
# per MI355X_MICROARCH.md "HBM": FETCH_SIZE reads exactly 1/2 of a wide coalesced stream on gfx950 -> doubled;
# unit KiB.  Writes gpurun_out/pmc_decode.json (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-2081}
[ -x $R/tools/mb/decode_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $R/tools/mb/decode_bench $R/tools/mb/decode_bench.cpp -L$R/spatten_amd/lib -lspatten_hip -Wl,-rpath,$R/spatten_amd/lib
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o p -- $R/tools/mb/decode_bench 1 $N 0 1 > /dev/null 2>&1
done
python3 - $R $N <<'PY'
import csv, glob, json, sys
R, N = sys.argv[1], int(sys.argv[2])
out = {"kernel": "decode_lean_kernel<bf16,128,5,...,512> (= decode_body, plain decode step, single-shot tile, 512-thread team)", "workload": f"B=1 H=32 d=128 kv_len={N} bf16, stash on (tools/mb/decode_bench)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if ("decode_attn_kernel" in r["Kernel_Name"] or "decode_lean_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == c]
    out[c + "_KiB_avg_per_launch"] = sum(v) / len(v)
    out[c + "_launches"] = len(v)
out["fetch_bytes_corrected_x2"] = out["FETCH_SIZE_KiB_avg_per_launch"] * 2 * 1024
out["write_bytes"] = out["WRITE_SIZE_KiB_avg_per_launch"] * 1024
out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = 2 * 32 * N * 128 * 2 + 2 * 32 * 128 * 2 + 32 * N * 2
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
json.dump(out, open(f"{R}/gpurun_out/pmc_decode.json", "w"), indent=1)
print(json.dumps(out))
PY
