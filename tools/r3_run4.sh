#!/bin/bash
# round-3 GPU call 4: 16-byte gathers in the long-row kernels -- tests, per-class timing, config-5 forward profile
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3e
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -5 "$out/pytest.log"
echo "== all classes (RMAT-20)"; timeout -k 10 300 python tools/exp_skew.py 2>/dev/null | tee "$out/skew_all.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rmat22" -- python3 "$root/tools/exp_train_step.py" --rmat 22 --edges 64000000 --feat 256 --forward-only > "$out/rmat22.log" 2>&1
grep "inference forward" "$out/rmat22.log"
cd "$root"; python tools/prof_top.py "$(find "$out/rmat22" -name "*kernel_stats.csv" | head -1)" 6 2>/dev/null | head -40
