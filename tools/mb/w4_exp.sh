# anatomy of the one-wave-per-SIMD prefill kernel: rebuild with -DSPATTEN_W4_EXP=<bits> (and -DSPATTEN_WITH_W4_EXPERIMENT -DSPATTEN_PF_TRACE), print phase cycles
#   bash tools/mb/w4_exp.sh 0 1 2 6 ...        (ARGS=fast for the fast-numerics instantiation)
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
mkdir -p /tmp/pft; rm -f /tmp/pft/*.o
for f in decode_attn prune cascade pq comm step gemv layer_cascade; do cp build/$f.o /tmp/pft/$f.o; done
for var in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -DSPATTEN_WITH_W4_EXPERIMENT -DSPATTEN_PF_TRACE -DSPATTEN_W4_EXP=$var $EXTRA -c spatten_amd/csrc/prefill_attn.hip -o /tmp/pft/prefill_attn.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pft/*.o -ldl
  echo "== SPATTEN_W4_EXP=$var $EXTRA $ARGS"; python tools/probe_w4_trace.py $ARGS 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
