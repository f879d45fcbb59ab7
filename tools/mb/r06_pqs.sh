cd $GRAFT_REPO_ROOT
for S in 0 12 16 17; do python tools/mb/pqv_exp.py 30 8192 $S 2>&1 | grep "8+4\|bf16 K"; done
