# q = N prefill under forced key split / pairing / transposing reads (developer A/B): bash tools/mb/pf_ks_sweep.sh
cd $GRAFT_REPO_ROOT
for N in 1024 2048 4096; do
for cfg in "" "SPATTEN_PREFILL_PAIR=0" "SPATTEN_PREFILL_PAIR=0 SPATTEN_PREFILL_KSPLIT=2" "SPATTEN_PREFILL_PAIR=0 SPATTEN_PREFILL_KSPLIT=2 SPATTEN_PREFILL_VTR=1" "SPATTEN_PREFILL_PAIR=0 SPATTEN_PREFILL_KSPLIT=2 SPATTEN_PREFILL_VTR=0" "SPATTEN_PREFILL_PAIR=0 SPATTEN_PREFILL_KSPLIT=4" "SPATTEN_PREFILL_PAIR=1 SPATTEN_PREFILL_KSPLIT=2"; do
  echo "N=$N [$cfg] $(env $cfg python tools/mb/pf2048_trace.py $N 2>&1 | tail -1)"
done; done
