cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
cp tools/mb/ab/lib_pftrace.so spatten_amd/lib/libspatten_hip.so
python tools/probe_pf_trace.py 2>&1 | grep -v amdgpu.ids
python tools/probe_pf_trace.py fast 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
