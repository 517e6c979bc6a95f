#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r4x
mkdir -p "$OUT"
cd "$ROOT"
FSV_EARLY_G=1 timeout 200 python -X faulthandler bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph > "$OUT/eager_1.tmp" 2> "$OUT/eager_1.err"
echo "eager rc=$?"; tail -n 1 "$OUT/eager_1.tmp" | cut -c1-200; tail -n 30 "$OUT/eager_1.err"
FSV_EARLY_G=1 timeout 200 python -X faulthandler bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/graph_1.tmp" 2> "$OUT/graph_1.err"
echo "graph rc=$?"; tail -n 1 "$OUT/graph_1.tmp" | cut -c1-200; tail -n 40 "$OUT/graph_1.err"
