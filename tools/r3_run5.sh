#!/bin/bash
# round-3 GPU call 5: mid class through LDS-DMA (FSW_MID_SPLIT=2), projection in two column groups (FSW_PROJECT_GROUPS=2)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3f
mkdir -p "$out"
cd "$root"
FSW_MID_SPLIT=2 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "mid_degree_rows_every or slice_blocks or homogeneity or hub_rows or wave_sort" > "$out/pytest_mid.log" 2>&1
echo "pytest (FSW_MID_SPLIT=2) rc=$?"; tail -3 "$out/pytest_mid.log"
FSW_PROJECT_GROUPS=2 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "projection or conv10k or er1m or fused" > "$out/pytest_proj.log" 2>&1
echo "pytest (FSW_PROJECT_GROUPS=2) rc=$?"; tail -3 "$out/pytest_proj.log"
echo "== mid class default"; timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_default.log"
echo "== mid class FSW_MID_SPLIT=2"; FSW_MID_SPLIT=2 timeout -k 10 300 python tools/exp_skew.py --fine --only mid 2>/dev/null | tee "$out/skew_lds.log"
echo "== bench default"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-segcumsum --no-weak --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
echo "== bench FSW_PROJECT_GROUPS=2"; FSW_PROJECT_GROUPS=2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-segcumsum --no-weak --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
echo "== rmat22 default"; timeout -k 10 400 python tools/exp_train_step.py --rmat 22 --edges 64000000 --feat 256 --forward-only 2>/dev/null | grep inference
echo "== rmat22 FSW_MID_SPLIT=2"; FSW_MID_SPLIT=2 timeout -k 10 400 python tools/exp_train_step.py --rmat 22 --edges 64000000 --feat 256 --forward-only 2>/dev/null | grep inference
