# P1 (reference roundings of S(t+1) behind the wave's own P.V MFMAs): tools/mb/ab/lib_P1.so is a -DSPATTEN_PF_P1=1 build
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in /tmp/lib_keep.so tools/mb/ab/lib_P1*.so; do cp $l spatten_amd/lib/libspatten_hip.so; for shape in "2048 2048" "4096 4096" "8192 8192"; do echo -n "$(basename $l)  "; timeout 60 python tools/probe_prefill_shape.py $shape 2>&1 | tail -1; done; done
cp tools/mb/ab/lib_P1.so spatten_amd/lib/libspatten_hip.so
timeout 600 python -m pytest tests/test_gpu_prefill.py -x -q -m gpu 2>&1 | tail -3
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
