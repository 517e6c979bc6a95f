#!/bin/bash
# round 3, fifth hardware pass: in-place A fragments on every tile shape (ids 16-18 new), per shape and whole step
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3e
mkdir -p "$OUT"
cd "$ROOT"
FSV_AB_EXPERIMENTAL=1 timeout 500 python tools/tile_ab.py "M131072 N128 K576" "M32768 N256 K1152" "M2048 N512 K2304" "M1024 N512 K512" "M524288 N32 K576" "M524288 N32 K288" "M131072 N32 K128" "M131072 N64 K288" > "$OUT/tile_ab.jsonl" 2> "$OUT/tile_ab.err"
AB_NAME=r3e_ab REPS=2 bash tools/hw_ab.sh base v4:FSV_CONV_V4=4 v2:FSV_CONV_V2=2 v0:FSV_CONV_V0=0 >> "$OUT/summary.txt" 2>&1
timeout 600 python -m pytest tests -q -m gpu -k "every_gemm_tile or conv_groups" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest tiles: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
cat "$OUT/summary.txt"
