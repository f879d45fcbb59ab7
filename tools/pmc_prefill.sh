# SQ counters of the prefill flash kernel (N = 8192 causal): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
timeout 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf2 -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
timeout 250 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pf3 -o p -- python $R/tools/probe_prefill.py > /dev/null 2>&1
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
# the q = N = 8192 launches of the flash kernel (grid 64 x 32 workgroups of 256 threads), grouped by instantiation: the probe runs the
# reference-numerics and the fp32-logit ("fast") instantiations back to back
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_pf", "pmc_pf2", "pmc_pf3"):
    fs = glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no output"); continue
    for r in csv.DictReader(open(fs[0])):
        if "prefill_pp128_kernel" in r["Kernel_Name"] and r["Grid_Size"] == str(64 * 32 * 256):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, cs in acc.items():
    print("kernel:", kn[:160])
    avg = {k: sum(v) / len(v) for k, v in cs.items()}
    if "SQ_INSTS_VALU" in avg and avg.get("SQ_INSTS_MFMA"):
        print(f"  SQ_INSTS_VALU / SQ_INSTS_MFMA = {avg['SQ_INSTS_VALU'] / 266240:.1f} / {avg['SQ_INSTS_MFMA'] / 266240:.1f} = {avg['SQ_INSTS_VALU'] / avg['SQ_INSTS_MFMA']:.2f}")
    for k, v in sorted(cs.items()):
        print(f"  {k:34s} {len(v):3d} {sum(v) / len(v):16.0f}  per wave-tile {sum(v) / len(v) / 266240:9.1f}")
PY
