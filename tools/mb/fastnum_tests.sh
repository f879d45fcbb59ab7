cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
mkdir -p /tmp/pfl; rm -f /tmp/pfl/*.o
for f in decode_attn prune cascade pq comm step gemv layer_cascade; do cp build/$f.o /tmp/pfl/$f.o 2>/dev/null || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -DSPATTEN_PF_FASTNUM=1 -c spatten_amd/csrc/prefill_attn.hip -o /tmp/pfl/prefill_attn.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pfl/*.o -ldl
python -m pytest tests/test_gpu_prefill.py tests/test_gpu_fullsize.py tests/test_gpu_cascade.py tests/test_gpu_e2e_protocol.py tests/test_gpu_random_sweep.py -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
