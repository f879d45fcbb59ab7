# phase stamps of the fused projection + attention launch: tools/mb/ab/lib_TRACE*.so are -DSPATTEN_TRACE -DSPATTEN_TRACE_SLOTS=16 builds
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in tools/mb/ab/lib_TRACE*.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/mb/fused_trace.py 2>&1 | grep layer; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
