# what bounds a step of the flash kernel at q = N = 2048 / 8192: stripped builds (WRONG results: timing only)
#   lib_PFEXP2 = no tile DMA inside the loop, lib_PFEXP8 = no softmax, lib_PFEXP10 = neither
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in /tmp/lib_keep.so tools/mb/ab/lib_PFEXP*.so; do cp $l spatten_amd/lib/libspatten_hip.so; for shape in "2048 2048" "8192 8192"; do echo -n "$(basename $l)  "; SPATTEN_PREFILL_PAIR=0 timeout 60 python tools/probe_prefill_shape.py $shape 2>&1 | tail -1; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
