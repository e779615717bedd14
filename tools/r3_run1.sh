#!/bin/bash
# round-3 GPU call 1: new tests, segcumsum variants, slice-shard compute + its kernel profile, self-launch rehearsal
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3b
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py -x -q -m gpu -k "segcumsum or legacy or slice_parallel or rccl or slice_blocks" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -3 "$out/pytest.log"
for v in new old nopf wg2; do
  lib=""; [ $v != new ] && lib=$root/_variants/libfsw_hip_seg_$v.so
  for e in 256000000 2560000000; do
    FSW_HIP_LIBRARY=$lib timeout -k 10 200 python tools/bench_segcumsum.py --elems $e --reps 5 --no-check 2>/dev/null | sed "s/^/$v $e /" >> "$out/segvariants.log" || echo "$v $e failed"
  done
done
FSW_HIP_LIBRARY= timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --ids i32 2>/dev/null | sed "s/^/new i32 /" >> "$out/segvariants.log"
FSW_HIP_LIBRARY= timeout -k 10 200 python tools/bench_segcumsum.py --elems 256000000 --dtype f64 --reverse 2>/dev/null | sed "s/^/new f64 rev /" >> "$out/segvariants.log"
cat "$out/segvariants.log"
timeout -k 10 300 python tools/exp_slice_shard.py > "$out/slice_shard.log" 2>&1; cat "$out/slice_shard.log"
timeout -k 10 300 python tools/exp_slice_shard.py --mode gather > "$out/slice_shard_gather.log" 2>&1; cat "$out/slice_shard_gather.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_w8" -- python3 "$root/tools/exp_slice_shard.py" --worlds 8 > "$out/prof_w8.log" 2>&1
f=$(find "$out/prof_w8" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
cd "$root"
FSW_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --kernel-reps 4 > "$out/bench_g2.json" 2> "$out/bench_g2.err"
echo "bench --gpus 2 (self-launched, gloo rehearsal) rc=$?"; tail -c 1500 "$out/bench_g2.json"; tail -5 "$out/bench_g2.err"
