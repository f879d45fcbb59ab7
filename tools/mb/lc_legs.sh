# layer-cascade prune event: legs of the chain with the gather of each leg on the side stream (SPATTEN_LC_LEGS, read once per process)
cd $GRAFT_REPO_ROOT
for n in 1 2 4 8 16; do echo "== SPATTEN_LC_LEGS=$n"; SPATTEN_LC_LEGS=$n timeout 120 python tools/probe_layer_cascade.py 2>&1 | tail -1; done
