# SQ counters of the prefill flash kernel (N = 8192 causal): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
timeout 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf2 -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
timeout 250 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf3 -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
for d in ("pmc_pf", "pmc_pf2", "pmc_pf3"):
    fs = glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "prefill_pp128_kernel" in r["Kernel_Name"] and ("Lb0ELi0ELb0ELb0ELb0EEEv" in r["Kernel_Name"] or "Lb0ELi0ELb0ELb0EEEv" in r["Kernel_Name"] or ", false, 0>" in r["Kernel_Name"]) and r["Grid_Size"] == str(64*32*256):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:34s} {len(v):3d} {sum(v) / len(v):16.0f}  per wave-tile {sum(v) / len(v) / 266240:9.1f}")
PY
