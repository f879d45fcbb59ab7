# A/B of two prebuilt libraries on the prefill probe: bash tools/mb/pf_ab_lib.sh /path/libA.so /path/libB.so   (alternates A B A B)
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for i in 1 2; do for l in "$@"; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/probe_prefill.py 2>&1 | grep -E "causal( fast)?:"; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
