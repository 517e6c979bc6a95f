#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2k
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
run pytest_ops   600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x
run bench_new    150 $B
run bench_nofuse 150 env FSV_SPADE_FUSED_BWD=0 $B
run bench_nofold 150 env FSV_SPADE_FOLD=0 $B
run bench_old    150 env FSV_SPADE_FUSED_BWD=0 FSV_SPADE_FOLD=0 $B
run bench_new2   150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
run pytest_c3    900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "c3 or c1" -rf
run pytest_model 900 python -m pytest tests/test_model_gpu.py tests/test_golden.py -q -m gpu -x
cat "$OUT/summary.txt"
