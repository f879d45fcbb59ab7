# A/B of prebuilt variant libraries (tools/mb/ab/lib_*.so: -DSPATTEN_FUSED_EXP=n builds of decode_attn.hip) on the layer-step probe
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in /tmp/lib_keep.so tools/mb/ab/lib_*.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/mb/fused_exp.py 2>&1 | grep rows | tail -1; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
