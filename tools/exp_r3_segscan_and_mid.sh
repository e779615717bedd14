#!/bin/bash
# round-3 GPU call 2: full GPU test suite on the new kernels, mid-class A/B, config-5 forward profile, segcumsum ablations
# Variants first (build container):  for a in 1 3 7; do tools/build_variant.sh seg_abl$a segcumsum "-DFSW_SEG_ABL=$a"; done;
#   tools/build_variant.sh seg_wg4np segcumsum "-DFSW_SEG_PREFETCH=0 -DFSW_SEG_WG_PER_CU=4"; tools/build_variant.sh midll2 embed_hub_0 "-DFSW_MIDSPLIT_LL=2";
#   tools/build_variant.sh midll8 embed_hub_0 "-DFSW_MIDSPLIT_LL=8"   (FSW_MID_SPLIT=1 selects the split kernels at run time)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3c
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$out/pytest.log"
for v in new abl1 abl3 abl7 wg4np; do
  lib=""; [ $v != new ] && lib=$root/_variants/libfsw_hip_seg_$v.so
  FSW_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --reps 5 --no-check 2>/dev/null | cut -c1-200 | sed "s/^/$v /" >> "$out/segvariants.log" || echo "$v failed"
done
cat "$out/segvariants.log"
echo "== mid class: split over 4 lanes (default)"; timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_split4.log"
echo "== mid class: one lane per slice (FSW_MID_SPLIT=0)"; FSW_MID_SPLIT=0 timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_nosplit.log"
echo "== LL=2"; FSW_HIP_LIBRARY=$root/_variants/libfsw_hip_midll2.so timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_split2.log"
echo "== LL=8"; FSW_HIP_LIBRARY=$root/_variants/libfsw_hip_midll8.so timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_split8.log"
echo "== all classes"; timeout -k 10 300 python tools/exp_skew.py 2>/dev/null | tee "$out/skew_all.log"
echo "== slice shard compute"; timeout -k 10 300 python tools/exp_slice_shard.py --worlds 8 2>/dev/null | tee "$out/slice_shard.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rmat22" -- python3 "$root/tools/exp_train_step.py" --rmat 22 --edges 64000000 --feat 256 --forward-only > "$out/rmat22.log" 2>&1
grep "inference forward" "$out/rmat22.log"
f=$(find "$out/rmat22" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "fsw::" "$f" | cut -d, -f1-4 | cut -c1-60,200- | head -30
