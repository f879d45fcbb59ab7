# decode kernel time when its K/V come from HBM (32 layers rotating: 1.1 GB) vs from the Infinity Cache (4 layers: 136 MB)
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/decode_bench tools/mb/decode_bench.cpp -Lspatten_amd/lib -lspatten_hip -Wl,-rpath,$GRAFT_REPO_ROOT/spatten_amd/lib
for n in 2048 2081; do for L in 32 8 4 2; do /tmp/decode_bench 1 $n 0 1 $L; done; done
