#!/bin/bash
# round-3 GPU call 10: narrow fused kernel at higher occupancy (6 / 8 waves per SIMD), slice-parallel tests
# Variants first (build container):  for w in 6 8; do tools/build_variant.sh fused_nw$w conv_fused "-DFSW_FUSED_NARROW_WAVES=$w"; done
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3k
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "slice_parallel or slice_blocks" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -3 "$out/pytest.log"
for v in default fused_nw6 fused_nw8; do
  lib=""; [ $v != default ] && lib=$root/_variants/libfsw_hip_$v.so
  echo "== $v"; FSW_HIP_LIBRARY=$lib timeout -k 10 300 python tools/exp_slice_shard.py --worlds 8 2>/dev/null | grep world
done
FSW_HIP_LIBRARY=$root/_variants/libfsw_hip_fused_nw8.so timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "slice_parallel or slice_blocks" 2>&1 | tail -2
