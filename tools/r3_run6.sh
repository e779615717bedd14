#!/bin/bash
# round-3 GPU call 6: gemm_tn tests, training step profile (config 3), backward tests
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3g
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "gemm_tn or backward or training or grads or slice_parallel" > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/pytest.log"
FSW_GEMM_TN_OFF=1 timeout -k 10 300 python tools/exp_train_step.py 2>/dev/null | tee "$out/train_blas.log"
timeout -k 10 300 python tools/exp_train_step.py 2>/dev/null | tee "$out/train_tn.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train" -- python3 "$root/tools/exp_train_step.py" > "$out/train_prof.log" 2>&1
cd "$root"; python tools/prof_top.py "$out/train" 22 | cut -c1-70,78-
