#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2j
mkdir -p "$OUT"
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
cd "$ROOT"
run bench        150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
run tile_ab      300 python tools/tile_ab.py "M4356 N256 K8192" "M4624 N512 K4096" "M2048 N512 K9216" "M8192 N256 K2304" "M512 N1024 K4608" "M2048 N512 K2304" "M8192 N256 K9216"
run bench2       150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline
cat "$OUT/summary.txt"
