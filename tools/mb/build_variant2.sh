# Like build_variant.sh, for a flag that lives in a shared header: EVERY listed unit is recompiled with the flags.
#   bash tools/mb/build_variant2.sh NAME "-DFLAG=1" decode_attn decode_chain      -> tools/mb/ab/lib_NAME.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2; shift 2
mkdir -p tools/mb/ab /tmp/var_$NAME
for UNIT in "$@"; do
  extra=""
  [ "$UNIT" = prefill_attn ] && extra="-fno-slp-vectorize"
  [ "$UNIT" = decode_attn ] && extra="-mllvm -amdgpu-kernarg-preload-count=16"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $extra $FLAGS -c spatten_amd/csrc/$UNIT.hip -o /tmp/var_$NAME/$UNIT.o &
done
wait
objs=""
for f in build/*.o; do b=$(basename $f .o); [ -f /tmp/var_$NAME/$b.o ] && objs="$objs /tmp/var_$NAME/$b.o" || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/mb/ab/lib_$NAME.so $objs -ldl
echo "built tools/mb/ab/lib_$NAME.so"
