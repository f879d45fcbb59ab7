# A/B harness for the select / local-V kernels: rebuild prune.hip (or the file named in F) with -D<macro>=<v> and time.
#   bash tools/mb/sel_exp.sh SPATTEN_SEL_EXP 0 1 2 3      (F=cascade for cascade.hip)
cd $GRAFT_REPO_ROOT
M=$1; shift
F=${F:-prune}
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
for var in "$@"; do
  mkdir -p /tmp/pfl; rm -f /tmp/pfl/*.o
  for f in decode_attn prefill_attn prune cascade pq comm step gemv layer_cascade; do
    if [ $f = $F ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -D$M=$var -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o &
    else cp build/$f.o /tmp/pfl/$f.o 2>/dev/null || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o &
    fi
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pfl/*.o -ldl
  echo "== $M=$var"; ${SCRIPT:-python tools/mb/localv_exp.py ${ARGS:-16384 40 0.3}} 2>&1 | tail -1
done
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
