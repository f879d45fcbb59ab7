# build the library with phase timestamps in the one-wave-per-SIMD prefill kernel, print the anatomy, restore the library
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
mkdir -p /tmp/pft; rm -f /tmp/pft/*.o
for f in decode_attn prune cascade pq comm step gemv layer_cascade; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c spatten_amd/csrc/$f.hip -o /tmp/pft/$f.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -DSPATTEN_WITH_W4_EXPERIMENT -DSPATTEN_PF_TRACE $EXTRA -c spatten_amd/csrc/prefill_attn.hip -o /tmp/pft/prefill_attn.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pft/*.o -ldl
python tools/probe_w4_trace.py $ARGS 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
