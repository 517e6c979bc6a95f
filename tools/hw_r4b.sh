#!/bin/bash
# round 4, pass b: the half-precision kernels on hardware for the first time - correctness against the definition, then the tile A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4b
mkdir -p "$OUT"
cd "$ROOT/tests"
timeout 300 python h_checks.py > "$OUT/h_checks.log" 2>&1
echo "h_checks: exit $? $(tail -n 1 "$OUT/h_checks.log" | cut -c1-300)" | tee -a "$OUT/summary.txt"
cd "$ROOT"
timeout 600 python tools/h_ab.py > "$OUT/h_ab.jsonl" 2> "$OUT/h_ab.err"
echo "h_ab: exit $?" | tee -a "$OUT/summary.txt"
cat "$OUT/h_ab.jsonl"
tail -5 "$OUT/h_ab.err"
