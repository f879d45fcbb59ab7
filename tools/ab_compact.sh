for u in 2 4 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DSPATTEN_COMP_UNROLL=$u -c spatten_amd/csrc/prune.hip -o build/prune.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so build/*.o
  echo "unroll=$u"; python bench.py --steps 64 --warmup 64 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['prune_event'])"
done
