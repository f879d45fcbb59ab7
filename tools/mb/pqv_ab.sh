# A/B of prebuilt variant libraries (tools/mb/ab/lib_*.so, built with -DSPATTEN_PQV_UP=n / -DSPATTEN_PQV_PVBAR=1) on the profiled-plane decode
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in /tmp/lib_keep.so tools/mb/ab/lib_*.so /tmp/lib_keep.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/mb/pqv_exp.py 32 8192 2>&1 | grep "V8\|V6"; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
