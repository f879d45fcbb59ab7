# A/B of prebuilt gemv variants (tools/mb/build_variant.sh NAME gemv "-D..."): bash tools/mb/gemv_ab.sh tools/mb/ab/lib_A.so ...
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for i in 1 2; do for l in /tmp/lib_keep.so "$@"; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/probe_graph_overheads.py 2>&1 | grep "gemv N"; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
