#!/bin/bash
# Round-2 hardware pass B: the rewritten fp32 gather-GEMM / weight-gradient kernels (buffer loads with hardware zero fill,
# double-buffered LDS, b128 A fragments, XCD-aware tile order) - parity on the GPU, per-shape A/B, whole step.
set -u
OUT=gpurun_out/r2b
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/summary.txt"
  local t0=$SECONDS
  timeout "$secs" "$@" > "$OUT/$name.log" 2>&1
  echo "   exit $? in $((SECONDS-t0))s ($(tail -n 1 "$OUT/$name.log" | cut -c1-300))" | tee -a "$OUT/summary.txt"
}
run pytest_ops   600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu
run tile_ab      200 python tools/tile_ab.py "M8192 N256 K2304" "M32768 N128 K1152" "M2048 N512 K2304" "M131072 N128 K576" "M512 N1024 K4608" "M131072 N64 K288"
run wgrad_ab     200 python tools/wgrad_ab.py
run bench_f32    200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run np_checks    300 python tests/np_checks.py
run pytest_model 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu
cat "$OUT/summary.txt"
