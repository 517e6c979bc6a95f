#!/bin/bash
# round 3, pass n: prefetch-distance-2 forms of the 4-wave tiles (ids 19 / 20) per shape; what the epilogue statistics cost per shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3n
mkdir -p "$OUT"
cd "$ROOT"
FSV_AB_EXPERIMENTAL=1 FSV_AB_TILES=2,4,17,18,19,20 FSV_AB_SPLITS=1 FSV_AB_STATS=1 timeout 400 python tools/tile_ab.py "M524288 N32 K576" "M524288 N32 K288" "M131072 N32 K128" "M131072 N64 K288" "M2048 N512 K2304" "M1024 N512 K512" "M32768 N64 K256" "M8192 N256 K2304" "M131072 N128 K576" > "$OUT/tile_ab.jsonl" 2> "$OUT/tile_ab.err"
echo "tile_ab exit $?" | tee -a "$OUT/summary.txt"
timeout 300 python -m pytest tests -q -m gpu -x -k "every_gemm_tile or layout_cache or deferred" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
AB_NAME=r3n_ab REPS=2 bash tools/hw_ab.sh base v2_19:FSV_CONV_V2=19 v4_20:FSV_CONV_V4=20 nostats:FSV_CONV_STATS=0 >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"; cat "$OUT/tile_ab.jsonl"
