#!/bin/bash
# Kernel tuning experiments: build variants of libfsw_hip.so with different -D flags (run in the build container),
# then time bench.py with each on the GPU box:  tools/exp_variants.sh build "name:-DFLAG=1 ..." ... | tools/exp_variants.sh run
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fsw_gnn_amd/csrc
out=$root/_variants
if [ "$1" = build ]; then
  shift
  mkdir -p "$out"
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    # the translation units that honour experiment flags are rebuilt, every other object comes from the regular build
    objs=""
    for o in $src/_build/*.o; do
      f=$(basename $o .o)
      case "$f" in
        embed_reg|conv_fused|project|graph_build|embed_wsort|embed_wsort_bwd) o=/tmp/var_${name}_$f.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include $flags -c $src/$f.hip -o $o ;;
        embed_hub_[0-2]) o=/tmp/var_${name}_$f.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include $flags -DFSW_HUB_PART=${f##*_} -c $src/embed_hub.hip -o $o ;;
        embed_mid|embed_lds|embed_hub) continue ;;   # stale objects of removed / split sources
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
    FSW_HIP_LIBRARY=$lib python "$root/bench.py" --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$out/$name.json" 2>"$out/$name.err" || { echo "$name FAILED"; tail -3 "$out/$name.err"; continue; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s step %.3f ms  %s" % (sys.argv[2], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}), flush=True)
PY
  done
fi
