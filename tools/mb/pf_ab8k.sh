# A/B of prebuilt libraries on q = N = 8192 only, four alternations: bash tools/mb/pf_ab8k.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for i in 1 2 3 4; do for l in /tmp/lib_keep.so "$@"; do cp $l spatten_amd/lib/libspatten_hip.so; echo -n "$(basename $l): "; python tools/probe_prefill_shape.py 8192 8192 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*call//'; echo -n "   fast: "; python tools/probe_prefill_shape.py 8192 8192 fast 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*call//'; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
