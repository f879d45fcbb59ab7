# A/B harness for the projection kernel: rebuild gemv.hip with -DSPATTEN_GEMV_ROWS=<r> and time it (tools/probe_graph_overheads.py)
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
for r in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DSPATTEN_GEMV_ROWS=$r -c spatten_amd/csrc/gemv.hip -o /tmp/gemv_exp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so $(ls build/*.o | grep -v gemv.o) /tmp/gemv_exp.o -ldl
  echo "== rows per wave $r"; python tools/probe_graph_overheads.py 2>&1 | grep "gemv N"
done
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
