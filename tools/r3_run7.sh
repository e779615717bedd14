#!/bin/bash
# round-3 GPU call 7: the whole GPU test suite, then the profile refresh (tools/refresh_profiles.sh r03v1)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3h
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$out/pytest.log"
bash tools/refresh_profiles.sh r03v1
