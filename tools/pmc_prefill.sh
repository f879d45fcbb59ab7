cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/pmc_pf/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "prefill_flash" in r["Kernel_Name"] and r["Grid_Size"] == str(64*32*256):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, len(v), sum(v) / len(v))
PY
