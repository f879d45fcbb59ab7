# build a traced copy of the library next to the microbenchmark and run it
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/trlib
for f in decode_attn prefill_attn prune cascade pq comm step gemv layer_cascade; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -mllvm -amdgpu-kernarg-preload-count=16 -DSPATTEN_TRACE -c spatten_amd/csrc/$f.hip -o /tmp/trlib/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/trlib/libspatten_hip.so /tmp/trlib/*.o -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/trlib/decode_trace tools/mb/decode_trace.cpp -L/tmp/trlib -lspatten_hip -Wl,-rpath,/tmp/trlib
/tmp/trlib/decode_trace 2048 8; /tmp/trlib/decode_trace 2081 8; /tmp/trlib/decode_trace 4096 8
