#!/bin/bash
# round 3, third hardware pass: mid-chunk-barrier tiles (A/B whole step + per shape), the tests that are new or failed in r3b
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3c
mkdir -p "$OUT"
cd "$ROOT"
AB_NAME=r3c_ab REPS=2 bash tools/hw_ab.sh base midbar:FSV_CONV_MIDBAR=1 >> "$OUT/summary.txt" 2>&1
FSV_AB_EXPERIMENTAL=1 timeout 300 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K2304" "M2048 N512 K2304" "M131072 N64 K288" > "$OUT/tile_ab.jsonl" 2> "$OUT/tile_ab.err"
t0=$SECONDS
timeout 1200 python -m pytest tests -q -m gpu -rf -k "test_c1 or test_c4 or test_c5 or epilogue or two_site or conv_groups or flownet or tile or prefetch" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest subset: exit $? in $((SECONDS-t0))s: $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
cat "$OUT/summary.txt"
