#!/bin/bash
# Round-2 hardware pass D: whole GPU suite (incl. the new full-size parity tests) on the 8-wave tiles / new plan, then A/B.
set -u
OUT=gpurun_out/r2d
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-200))" | tee -a "$OUT/summary.txt"
}
run pytest_full  900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu --durations=5
run bench        150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_w8     150 env FSV_WGRAD_8W=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run tile_ab      300 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M2048 N512 K2304" "M131072 N128 K576" "M512 N1024 K4608" "M131072 N64 K288" "M8192 N128 K512" "M32768 N64 K256"
run wgrad_ab     300 python tools/wgrad_ab.py
run pytest_gpu   1200 python -m pytest tests -x -q -m gpu --deselect tests/test_fullsize_gpu.py --durations=8
run bench2       150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
cat "$OUT/summary.txt"
