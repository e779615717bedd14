#!/bin/bash
# round-3 GPU call 8: the whole GPU test suite after the in-place fix of the segmented scan, smoke(), one bench line
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3i
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -q -m gpu > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -6 "$out/pytest.log"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline > "$out/bench.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'], d['segcumsum']['achieved'])"
timeout -k 10 200 python tools/bench_segcumsum.py --elems 2560000000 --reps 3 2>/dev/null | cut -c80-300
