# HBM traffic of the c5 layer-step (pqv_decode_kernel: MSB pass + LSB refetch pass) from PMC counters — tools/pmc_bench.sh's
# procedure on tools/mb/c5_step.py (the bench command itself crashes inside rocprofv3 --pmc at this geometry).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcc5_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcc5_$c -o p -- python $R/tools/mb/c5_step.py ${1:-trace} > $R/gpurun_out/pmcc5_$c.log 2>&1
done
python3 - $R <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
info = json.loads([l for l in open(f"{R}/gpurun_out/pmcc5_FETCH_SIZE.log") if l.startswith("C5_STEP_JSON ")][-1][len("C5_STEP_JSON "):])
out = {"config": "c5", "launch": "per-layer", "kernels_matched": ["pqv_decode_kernel"],
       "command": "tools/mb/c5_step.py (4 layers x 24 steps of the c5 layer-step; bench.py --config c5 itself crashes inside rocprofv3 --pmc)", **info}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmcc5_{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "pqv_decode_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KiB_sum"] = sum(v)
    out[c + "_launches"] = len(v)
steps = info["layer_steps"]
out["fetch_bytes_corrected_x2"] = out["FETCH_SIZE_KiB_sum"] / steps * 2 * 1024
out["write_bytes"] = out["WRITE_SIZE_KiB_sum"] / steps * 1024
out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = info["algorithmic_bytes_per_layer_step"]
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
json.dump(out, open(f"{R}/gpurun_out/pmc_decode_c5.json", "w"), indent=1)
json.dump(out, open(f"{R}/gpurun_out/pmc_decode_c5_{info['pq_confidence']}.json", "w"), indent=1)
print(json.dumps(out))
PY
