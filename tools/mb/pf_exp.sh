cd $GRAFT_REPO_ROOT
for var in 0 1 2 3 4; do
  mkdir -p /tmp/pfl; rm -f /tmp/pfl/*.o
  for f in decode_attn prefill_attn prune cascade pq; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DSPATTEN_PF_SCHED=$var -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pfl/*.o
  echo "== sched $var"; python tools/probe_prefill.py 2>&1 | grep "causal:"
done
