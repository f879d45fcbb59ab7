# phase stamps of the flash kernel from PREBUILT trace libraries (tools/mb/build_variant.sh NAME prefill_attn "-DSPATTEN_PF_TRACE ..."):
#   bash tools/mb/pf_trace_prebuilt.sh tools/mb/ab/lib_pftrace.so [more libs]
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in "$@"; do
  cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"
  python tools/probe_pf_trace.py 2>&1 | grep -v amdgpu.ids
  python tools/probe_pf_trace.py fast 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
