# A/B harness for the decode kernel: rebuild decode_attn.hip with extra flags and run the bench (no extras)
#   bash tools/mb/dec_exp.sh "" "-DSPATTEN_DECODE_NT"        (CMD="python tools/mb/pq_exp.py" times something else)
cd $GRAFT_REPO_ROOT
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
mkdir -p /tmp/dex; rm -f /tmp/dex/*.o
for f in prune cascade pq comm step gemv layer_cascade; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -c spatten_amd/csrc/$f.hip -o /tmp/dex/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-slp-vectorize -c spatten_amd/csrc/prefill_attn.hip -o /tmp/dex/prefill_attn.o &
wait
for var in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -mllvm -amdgpu-kernarg-preload-count=16 $var -c spatten_amd/csrc/decode_attn.hip -o /tmp/dex/decode_attn.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/dex/*.o -ldl
  echo "== flags: [$var]"
  for i in 1 2; do
    if [ -n "$CMD" ]; then eval "$CMD" 2>&1 | tail -1
    else python bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])"
    fi
  done
done
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
