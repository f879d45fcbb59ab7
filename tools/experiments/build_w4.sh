# Experiment build of the library WITH the one-wave-per-SIMD prefill kernel (tools/experiments/prefill_w4.h, round 3: parity
# green, slower than the product kernel), run its parity check, restore the product library.
#   bash tools/experiments/build_w4.sh            (on the GPU box; EXTRA=-D... for the anatomy macros)
cd ${GRAFT_REPO_ROOT:-.}
cp spatten_amd/lib/libspatten_hip.so /tmp/lib_orig.so
mkdir -p /tmp/w4b; rm -f /tmp/w4b/*.o
for f in spatten_amd/csrc/*.hip; do
  n=$(basename $f .hip); fl=""
  [ $n = prefill_attn ] && fl="-fno-slp-vectorize -DSPATTEN_WITH_W4_EXPERIMENT $EXTRA"
  [ $n = decode_attn ] && fl="-mllvm -amdgpu-kernarg-preload-count=16"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $fl -c $f -o /tmp/w4b/$n.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o spatten_amd/lib/libspatten_hip.so /tmp/w4b/*.o -ldl
python -m pytest tools/experiments/check_prefill_w4.py -q -m gpu 2>&1 | tail -3
${AFTER:-true}
cp /tmp/lib_orig.so spatten_amd/lib/libspatten_hip.so
