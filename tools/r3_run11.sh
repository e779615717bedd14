#!/bin/bash
# round-3 GPU call 11: training step on a skewed graph (RMAT-20) -- where the backward of long rows stands
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3l
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train20" -- python3 "$root/tools/exp_train_step.py" --rmat 20 > "$out/train20.log" 2>&1
grep -E "inference|training" "$out/train20.log"
cd "$root"; python tools/prof_top.py "$out/train20" 25 | cut -c1-74,78-
