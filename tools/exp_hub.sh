#!/bin/bash
# Ablation builds of the long-row kernels (run in the build container): tools/exp_hub.sh build "name:-DFLAG=1" ...
# then on the GPU box: tools/exp_hub.sh run [exp_skew.py arguments]  -- times every class with every variant library.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fsw_gnn_amd/csrc
out=$root/_variants
if [ "$1" = build ]; then
  shift
  mkdir -p "$out"
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    objs=""
    for o in $src/_build/*.o; do
      f=$(basename $o .o)
      case "$f" in
        embed_hub|embed_wsort) o=/tmp/varh_${name}_$f.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include $flags -c $src/$f.hip -o $o ;;
      esac
      objs="$objs $o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libfsw_hip_$name.so
    echo "built $name"
  done
else
  [ "$1" = run ] && shift
  for lib in "$out"/libfsw_hip_*.so; do
    name=$(basename "$lib" .so); name=${name#libfsw_hip_}
    echo "== $name"
    FSW_HIP_LIBRARY=$lib python "$root/tools/exp_skew.py" "$@" 2>&1 | grep -v amdgpu.ids
  done
fi
