# ping-pong offset vs lock-step halves, with / without P1 (tools/mb/ab/lib_PF_L{0,1}P{0,1}.so: -DSPATTEN_PF_LOCKSTEP / -DSPATTEN_PF_P1 builds)
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for l in /tmp/lib_keep.so tools/mb/ab/lib_PF_L*.so; do cp $l spatten_amd/lib/libspatten_hip.so; for shape in "2048 2048" "8192 8192"; do echo -n "$(basename $l)  "; timeout 60 python tools/probe_prefill_shape.py $shape 2>&1 | tail -1; done
  if [ $l != /tmp/lib_keep.so ]; then timeout 300 python -m pytest tests/test_gpu_prefill.py -x -q -m gpu -k "oracle or goldens" 2>&1 | tail -1; fi; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
