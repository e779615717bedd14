#!/bin/bash
# round-3 GPU call 12: quad-structured backward for 129..2048 neighbours: tests, RMAT-20 training step with and without it
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3m
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "backward or grads or training or store_and_sum or readout" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/pytest.log"
echo "== old backward kernels"; FSW_BWD_QUAD_OFF=1 timeout -k 10 300 python tools/exp_train_step.py --rmat 20 2>/dev/null | grep -E "training"
echo "== quad backward kernels"; timeout -k 10 300 python tools/exp_train_step.py --rmat 20 2>/dev/null | grep -E "training"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train20" -- python3 "$root/tools/exp_train_step.py" --rmat 20 > "$out/train20.log" 2>&1
cd "$root"; python tools/prof_top.py "$out/train20" 14 | cut -c1-74,78-
