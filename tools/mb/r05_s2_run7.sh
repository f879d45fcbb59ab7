cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in tools/mb/ab/lib_pfdmai.so tools/mb/ab/lib_pfdmai_rs_nm.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== parity $l"; timeout 600 python -m pytest tests/test_gpu_prefill.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
bash tools/mb/pf_ab_lib.sh /tmp/lib_keep.so tools/mb/ab/lib_pfdmai.so tools/mb/ab/lib_pfdmai_rs.so tools/mb/ab/lib_pfdmai_nm.so tools/mb/ab/lib_pfdmai_rs_nm.so
