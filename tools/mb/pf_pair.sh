# paired 128-row blocks of the flash kernel (SPATTEN_PREFILL_PAIR: 0 off, 1 forced, 2 auto), per shape
cd $GRAFT_REPO_ROOT
for shape in "1024 1024" "2048 2048" "2048 4096" "4096 4096" "8192 8192"; do
  for pm in 0 1; do echo -n "PAIR=$pm  "; SPATTEN_PREFILL_PAIR=$pm timeout 120 python tools/probe_prefill_shape.py $shape 2>&1 | tail -1; done
done
