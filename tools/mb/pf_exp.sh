# A/B harness for the prefill hot path: rebuild the library with -D<macro>=<v> for each value and time causal prefill.
#   bash tools/mb/pf_exp.sh SPATTEN_PF_PACK 0 1 2
cd $GRAFT_REPO_ROOT
M=$1; shift
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
for var in "$@"; do
  mkdir -p /tmp/pfl; rm -f /tmp/pfl/*.o
  for f in $(ls spatten_amd/csrc/*.hip | xargs -n1 basename | sed 's/.hip//'); do
    if [ $f = prefill_attn ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -D$M=$var -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o &
    else cp build/$f.o /tmp/pfl/$f.o 2>/dev/null || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c spatten_amd/csrc/$f.hip -o /tmp/pfl/$f.o &
    fi
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/pfl/*.o -ldl
  echo "== $M=$var"; python tools/probe_prefill.py 2>&1 | grep "causal:"
done
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
