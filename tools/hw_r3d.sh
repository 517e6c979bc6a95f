#!/bin/bash
# round 3, fourth hardware pass: in-place A fragments (tile ids 13-15) per shape and on the whole step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3d
mkdir -p "$OUT"
cd "$ROOT"
FSV_AB_EXPERIMENTAL=1 timeout 400 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K2304" "M32768 N256 K1152" "M131072 N64 K288" "M131072 N128 K576" > "$OUT/tile_ab.jsonl" 2> "$OUT/tile_ab.err"
AB_NAME=r3d_ab REPS=2 bash tools/hw_ab.sh base af:FSV_CONV_AF=1 >> "$OUT/summary.txt" 2>&1
timeout 600 python -m pytest tests -q -m gpu -k "every_gemm_tile" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest tiles: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
cat "$OUT/summary.txt"
