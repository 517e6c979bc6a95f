#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2m
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
run bench_off   150 env FSV_BRANCH_STREAMS=0 $B
run bench_on    150 $B
run bench_off2  150 env FSV_BRANCH_STREAMS=0 $B
run bench_on2   150 $B
run graph_step  300 env PYTHONPATH=tests python tests/graph_step_checks.py
run pytest_model 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_golden.py -q -m gpu -k "not c2" -rf
cat "$OUT/summary.txt"
