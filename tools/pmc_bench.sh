# HBM traffic of the dominant kernel of `bench.py --config C` from PMC counters: two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, per MI355X_MICROARCH.md "HBM" (FETCH_SIZE reads exactly 1/2 of a wide
# coalesced stream on gfx950 -> doubled; unit KiB).  Writes gpurun_out/pmc_decode_<config>.json (copy into profiles/ as
# rNN_pmc_decode_<config>.json: bench.py reads the newest *pmc_decode_<config>.json for `roofline.traffic`).
#   bash tools/pmc_bench.sh c2 [extra bench flags]        (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-c2}; shift
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcb_${C}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_${C}_$c -o p -- \
    python $R/bench.py --config $C --steps 24 --warmup 40 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/pmcb_${C}_$c.log 2>&1
done
python3 - $R $C <<'PY'
import csv, glob, json, sys
R, C = sys.argv[1], sys.argv[2]
line = json.loads([l for l in open(f"{R}/gpurun_out/pmcb_{C}_FETCH_SIZE.log") if l.startswith("{")][-1])
chained = "chained" in line["config"]["launch"]
# the kernels of a layer-step: the chained launch (a token), the per-layer decode launch, or the two passes of the quantised step
pats = ["decode_chain_kernel"] if chained else (["pqv_decode_kernel"] if C == "c5" else ["decode_lean_kernel", "decode_lean_hids_kernel", "decode_attn_kernel"])
out = {"config": C, "launch": "chained" if chained else "per-layer", "kernels_matched": pats,
       "command": f"bench.py --config {C} --steps 24 --warmup 40 --no-cpu-baseline --no-extras (under rocprofv3 --pmc, one counter per pass)"}
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmcb_{C}_{c}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for p in pats) and r["Counter_Name"] == c]
    # skip the set-up launches (the dense 4096-row steps that build the stash) by taking the steady tail
    v = v[len(v) // 4:]
    tot[c] = (sum(v), len(v))
    out[c + "_KiB_sum"] = sum(v)
    out[c + "_launches"] = len(v)
rf = line["roofline"]
layers = line["config"]["layers"]
# per `roofline` launch: the chained kernel is one launch per token; the per-layer forms are normalised to a LAYER-STEP (c5: the MSB
# pass + the refetch pass of a layer are two launches of one step)
steps = tot["FETCH_SIZE"][1] if chained else None
if chained:
    per = lambda c: tot[c][0] / tot[c][1]
else:
    n_steps = tot["FETCH_SIZE"][1] / (2 if C == "c5" else 1)
    per = lambda c: tot[c][0] / n_steps
out["fetch_bytes_corrected_x2"] = per("FETCH_SIZE") * 2 * 1024
out["write_bytes"] = per("WRITE_SIZE") * 1024
out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = rf["algorithmic_bytes_per_launch"]
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / rf["algorithmic_bytes_per_launch"]
out["bench_line_under_rocprof"] = {k: line[k] for k in ("value", "ms_per_step")}
json.dump(out, open(f"{R}/gpurun_out/pmc_decode_{C}.json", "w"), indent=1)
print(json.dumps(out))
PY
