#!/bin/bash
# Round-2 hardware pass C: fragment pipeline on / off, 8-wave workgroups - per-shape and whole-step A/B in one box.
set -u
OUT=gpurun_out/r2c
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
run tiles        300 python tests/tile_checks.py
run tile_ab      300 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M2048 N512 K2304" "M131072 N128 K576" "M512 N1024 K4608" "M131072 N64 K288" "M32768 N256 K1152" "M8192 N256 K1152"
run bench_pipe   150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_nopipe 150 env FSV_TILE_MAP=0:20,1:21,2:22,4:24,9:29 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_8w     150 env FSV_TILE_MAP=0:10,1:11,9:12 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_8w9    150 env FSV_TILE_MAP=9:12 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_pipe2  150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
cat "$OUT/summary.txt"
