#!/bin/bash
# Runs on the GPU box (gpurun -- tools/refresh_profiles.sh TAG): the default bench line, the rocprofv3 kernel statistics of
# the same command, the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only, as the MI355X guide
# prescribes), the kernel statistics of the skewed-degree forward (BASELINE config 5's shape on one GPU) and of the stand-alone
# segmented cumsum.  Outputs land in gpurun_out/refresh_TAG/; tools/collect_profiles.py copies the summaries into profiles/.
set -o pipefail
tag=${1:-latest}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/refresh_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$root/bench.py" > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
tail -c 600 "$out/bench.json"; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-weak > "$out/stats.log" 2>&1 || { echo "stats run failed"; tail -5 "$out/stats.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" --steps 2 --warmup 1 --kernel-reps 3 --no-cpu-baseline --no-segcumsum --no-weak > "$out/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$out/pmc_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" --steps 2 --warmup 1 --kernel-reps 3 --no-cpu-baseline --no-segcumsum --no-weak > "$out/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$out/pmc_write.log"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rmat22" -- python3 "$root/tools/exp_train_step.py" --rmat 22 --edges 64000000 --feat 256 --forward-only > "$out/rmat22.log" 2>&1 || { echo "rmat22 run failed"; tail -5 "$out/rmat22.log"; exit 1; }
grep "inference forward" "$out/rmat22.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/segcumsum" -- python3 "$root/tools/bench_segcumsum.py" --elems 2560000000 --reps 3 --no-check > "$out/segcumsum.log" 2>&1 || { echo "segcumsum run failed"; tail -5 "$out/segcumsum.log"; exit 1; }
grep "^{" "$out/segcumsum.log" | tail -1 > "$out/segcumsum.json"; cat "$out/segcumsum.json"
# HBM traffic of the scan (separate PMC passes, kernel trace only): 2.56e8 elements
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/seg_pmc_fetch" -- python3 "$root/tools/bench_segcumsum.py" --elems 256000000 --reps 2 --no-check > "$out/seg_pmc_fetch.log" 2>&1 || echo "segcumsum fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/seg_pmc_write" -- python3 "$root/tools/bench_segcumsum.py" --elems 256000000 --reps 2 --no-check > "$out/seg_pmc_write.log" 2>&1 || echo "segcumsum write pass failed"
timeout -k 10 300 python3 "$root/tools/exp_skew.py" > "$out/skew_rmat20.log" 2>/dev/null; cat "$out/skew_rmat20.log"
timeout -k 10 300 python3 "$root/tools/exp_slice_shard.py" --worlds 4,8 > "$out/slice_shard_consumer.log" 2>/dev/null; cat "$out/slice_shard_consumer.log"
timeout -k 10 300 python3 "$root/tools/exp_slice_shard.py" --worlds 4,8 --mode exchange > "$out/slice_shard_exchange.log" 2>/dev/null; cat "$out/slice_shard_exchange.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train" -- python3 "$root/tools/exp_train_step.py" > "$out/train.log" 2>&1 || echo "train step profile failed"
grep "training step" "$out/train.log"
find "$out" -name "*kernel_stats.csv" | head -20
