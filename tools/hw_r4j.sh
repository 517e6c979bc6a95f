#!/bin/bash
# round 4, pass j: the full-size --amp parity test (C5) on hardware + the half-kernel checks
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-r4j}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "c5_street_1024x512_nc35_amp" -s > "$OUT/pytest_c5_amp.log" 2>&1
echo "c5 amp: exit $? $(grep 'amp step' "$OUT/pytest_c5_amp.log" | cut -c1-400)" | tee -a "$OUT/summary.txt"
tail -n 5 "$OUT/pytest_c5_amp.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
