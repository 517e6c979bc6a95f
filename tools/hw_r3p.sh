#!/bin/bash
# round 3, pass p: offset arithmetic behind the first MFMA group (forward / data-gradient / weight-gradient loops): knock-outs again,
# tiles per shape, whole step against the previous build (tools/_diag/libfsv2v_hip_base.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r3p
mkdir -p "$OUT"
cd "$ROOT"
FSV2V_LIB=$ROOT/tools/_diag/libfsv2v_hip_diag.so timeout 300 python tools/knockout.py > "$OUT/knockout.jsonl" 2> "$OUT/knockout.err"
timeout 400 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K2304" "M131072 N128 K576" "M32768 N256 K1152" "M2048 N512 K2304" "M524288 N32 K576" "M524288 N32 K288" "M131072 N64 K288" > "$OUT/tile_ab_new.jsonl" 2> "$OUT/tile_ab.err"
FSV2V_LIB=$ROOT/tools/_diag/libfsv2v_hip_base.so timeout 400 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K2304" "M131072 N128 K576" "M32768 N256 K1152" "M2048 N512 K2304" "M524288 N32 K576" "M524288 N32 K288" "M131072 N64 K288" > "$OUT/tile_ab_base.jsonl" 2>> "$OUT/tile_ab.err"
timeout 300 python tools/wgrad_ab.py > "$OUT/wgrad_ab_new.jsonl" 2> "$OUT/wgrad_ab.err"
FSV2V_LIB=$ROOT/tools/_diag/libfsv2v_hip_base.so timeout 300 python tools/wgrad_ab.py > "$OUT/wgrad_ab_base.jsonl" 2>> "$OUT/wgrad_ab.err"
timeout 300 python -m pytest tests -q -m gpu -x -k "every_gemm_tile or conv_groups or test_conv" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest: exit $? $(tail -n 1 "$OUT/pytest_gpu.log")" | tee -a "$OUT/summary.txt"
AB_NAME=r3p_ab REPS=2 bash tools/hw_ab.sh new base:FSV2V_LIB=$ROOT/tools/_diag/libfsv2v_hip_base.so >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
