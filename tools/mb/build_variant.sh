# Build a variant of the library HERE (no GPU needed) for an A/B on the GPU box: one translation unit recompiled with extra
# flags, the rest taken from build/ (run `make -j8 lib` first).  The .so travels with the snapshot (git-ignored).
#   bash tools/mb/build_variant.sh NAME prefill_attn "-DSPATTEN_PF_DIET=0"      -> tools/mb/ab/lib_NAME.so
#   bash tools/mb/pf_ab_lib.sh tools/mb/ab/lib_A.so tools/mb/ab/lib_B.so        (on the box)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; UNIT=$2; FLAGS=$3
mkdir -p tools/mb/ab /tmp/var_$NAME
extra=""
[ "$UNIT" = prefill_attn ] && extra="-fno-slp-vectorize"
[ "$UNIT" = decode_attn ] && extra="-mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $extra $FLAGS -c spatten_amd/csrc/$UNIT.hip -o /tmp/var_$NAME/$UNIT.o
objs=""
for f in build/*.o; do b=$(basename $f .o); [ "$b" = "$UNIT" ] && objs="$objs /tmp/var_$NAME/$UNIT.o" || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/ab/lib_$NAME.so $objs -ldl
echo "built tools/mb/ab/lib_$NAME.so"
