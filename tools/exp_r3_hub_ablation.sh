#!/bin/bash
# round-3 GPU call 3: exchange form + halo segscan tests, segscan variants, hub ablations, slice-shard compute, bench lines
# Variants first (build container):  tools/build_variant.sh hub_nogather embed_hub_0 "-DFSW_HUB_ABL=1"; tools/build_variant.sh hub_gatheronly
#   embed_hub_0 "-DFSW_HUB_ABL=14"; tools/build_variant.sh hub_noreadout embed_hub_0 "-DFSW_HUB_ABL=8"; seg_* as in the segscan scripts
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3d
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "segcumsum or legacy or slice_parallel or rccl" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/pytest.log"
for v in new np np4 nohalo abl3 np_abl3; do
  lib=""; [ $v != new ] && lib=$root/_variants/libfsw_hip_seg_$v.so
  FSW_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --reps 5 --no-check 2>/dev/null | cut -c90-200 | sed "s/^/$v 2.56e8 /" >> "$out/segvariants.log" || echo "$v failed"
done
for v in new np; do
  lib=""; [ $v != new ] && lib=$root/_variants/libfsw_hip_seg_$v.so
  FSW_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_segcumsum.py --elems 2560000000 --reps 3 2>/dev/null | cut -c90-260 | sed "s/^/$v 2.56e9 /" >> "$out/segvariants.log" || echo "$v failed"
done
FSW_HIP_LIBRARY= timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --mean-seg 5000 2>/dev/null | cut -c90-260 | sed "s/^/new long-segments /" >> "$out/segvariants.log"
FSW_HIP_LIBRARY= timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --reverse --ids i32 2>/dev/null | cut -c90-260 | sed "s/^/new rev i32 /" >> "$out/segvariants.log"
cat "$out/segvariants.log"
echo "== hub ablations"
for v in default hub_nogather hub_gatheronly hub_noreadout; do
  lib=""; [ $v != default ] && lib=$root/_variants/libfsw_hip_$v.so
  echo "-- $v"; FSW_HIP_LIBRARY=$lib timeout -k 10 300 python tools/exp_skew.py 2>/dev/null | grep -E "ws|hub|global" | tee -a "$out/hub_abl_$v.log"
done
echo "== slice shard compute"
timeout -k 10 300 python tools/exp_slice_shard.py --worlds 4,8 2>/dev/null | tee "$out/slice_shard.log"
timeout -k 10 300 python tools/exp_slice_shard.py --worlds 4,8 --mode exchange 2>/dev/null | tee "$out/slice_shard_x.log"
echo "== bench N=1"
timeout -k 10 400 python bench.py > "$out/bench1.json" 2> "$out/bench1.err"; echo "rc=$?"; tail -c 1800 "$out/bench1.json"
echo "== bench --gpus 2 (gloo rehearsal)"
FSW_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --kernel-reps 4 > "$out/bench_g2.json" 2> "$out/bench_g2.err"; echo "rc=$?"; tail -c 2500 "$out/bench_g2.json"; tail -3 "$out/bench_g2.err"
