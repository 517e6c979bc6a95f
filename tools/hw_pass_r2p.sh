#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2p
mkdir -p "$OUT"
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
cd "$ROOT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline"
run a_off    150 env FSV_BRANCH_STREAMS=0 $B
run b_base   150 $B
run c_lemb   150 env FSV_BRANCH_TAGS=lemb $B
run d_base   150 $B
run e_lemb   150 env FSV_BRANCH_TAGS=lemb $B
grep -o '"ms_per_step": [0-9.]*' "$OUT"/*.log
