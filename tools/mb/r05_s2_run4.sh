cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pq_profiles.py tests/test_gpu_decode.py tests/test_gpu_prefill.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_keep.so
for i in 1 2; do for l in /tmp/lib_keep.so tools/mb/ab/lib_pqvold.so; do cp $l spatten_amd/lib/libspatten_hip.so; echo "== $l"; python tools/mb/pqv_exp.py 32 8192 2>&1 | grep -v amdgpu; done; done
cp /tmp/lib_keep.so spatten_amd/lib/libspatten_hip.so
for shape in "1024 1024" "1536 1536" "2048 2048" "2560 2560" "3072 3072" "1024 2048" "2048 4096"; do for v in 0 1; do echo -n "VTR=$v $shape: "; SPATTEN_PREFILL_VTR=$v python tools/probe_prefill_shape.py $shape 2>&1 | grep -v amdgpu | tail -1; done; done
