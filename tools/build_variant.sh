#!/bin/bash
# tools/build_variant.sh NAME UNIT "FLAGS" : _variants/libfsw_hip_NAME.so = the regular build with translation unit UNIT
# (segcumsum | embed_hub_0 | conv_fused | ...) recompiled with FLAGS.  Run in the build container after `make`; select the
# library on the GPU box with FSW_HIP_LIBRARY=_variants/libfsw_hip_NAME.so (fsw_gnn_amd/_lib.py).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fsw_gnn_amd/csrc
name=$1; unit=$2; flags=$3
mkdir -p "$root/_variants" /tmp/fsw_variants
file=$unit; extra=""
case "$unit" in
  embed_hub_[0-2]) file=embed_hub; extra="-DFSW_HUB_PART=${unit##*_}" ;;
  embed_mid_[0-2]) file=embed_mid; extra="-DFSW_MID_PART=${unit##*_}" ;;
  embed_mid_bwd_[0-1]) file=embed_mid_bwd; extra="-DFSW_MID_BWD_PART=${unit##*_}" ;;
esac
obj=/tmp/fsw_variants/${name}_$unit.o
(cd "$src" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra $flags -c $file.hip -o $obj)
objs=$(ls $src/_build/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o "$root/_variants/libfsw_hip_$name.so"
echo "built _variants/libfsw_hip_$name.so"
