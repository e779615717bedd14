#!/bin/bash
# round-3 GPU call 9: segscan register-cap variants (workgroups per CU), segcumsum tests, d = 256 fused path check (yin by node)
# Variants first (build container):  for w in 4 5 6; do tools/build_variant.sh seg_mw$w segcumsum "-DFSW_SEG_MINWAVES=$w"; done
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3j
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "segcumsum or fused or few_long or node" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -3 "$out/pytest.log"
for v in default seg_mw4 seg_mw5 seg_mw6; do
  lib=""; [ $v != default ] && lib=$root/_variants/libfsw_hip_$v.so
  for e in 256000000 2560000000; do
    FSW_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_segcumsum.py --elems $e --reps 4 --no-check 2>/dev/null | cut -c80-200 | sed "s/^/$v $e /"
  done
done
echo "== rmat22 forward (yin by node)"; timeout -k 10 400 python tools/exp_train_step.py --rmat 22 --edges 64000000 --feat 256 --forward-only 2>/dev/null | grep inference
